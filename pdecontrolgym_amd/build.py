"""Build recipe for libpdegym_hip.so (hand-written HIP for gfx950, C ABI in include/pdegym.h).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so stays
in-tree (git-ignored) and travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import hashlib
import re
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpdegym_hip.so")
SOURCES = ["pdegym_abi.hip", "pdegym_1d.hip", "pdegym_1d_rollout.hip", "pdegym_ns2d.hip", "pdegym_ns256.hip", "pdegym_ns256_f64.hip", "pdegym_traffic.hip", "pdegym_tumor.hip", "pdegym_mlp.hip"]
# -ffp-contract=off: NumPy rounds after every operation; a fused multiply-add would break bit parity.
# -fno-slp-vectorize: v_pk_*_f32 has the same per-element issue cost as the scalar forms on gfx950 (tools/ubench_valu.hip:
# 5.1 vs 2.8 cycles per wave-instruction at 4 waves/SIMD) and packing adjacent stencil nodes costs shuffle moves.
# -falign-loops=32 (round 5): loop headers on 32-byte boundaries.  The headline kernel's 26-instruction sub-step loop is identical in
# two builds that differ only in where it starts, and runs 4 % apart (16.3 against 17.0 us per launch; any of 32 / 64 / 128 gives the
# quicker figure, every other workload moves by < 1 %: profiles/r05_ab_notes.txt section 8) -- instruction fetch of a tight loop
# depends on how many fetch windows its body straddles, so the placement is pinned instead of left to what precedes the loop.
OPTS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-falign-loops=32", "-fPIC"]
FLAGS = OPTS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _fingerprint() -> str:
    """Hash of what the library is built from: the sources, the public header and the compiler options -- no absolute paths
    (the include directories are named by the files hashed above), so the stamp survives moving or copying the tree."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/pdegym.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    h.update(" ".join(OPTS).encode())
    return h.hexdigest()


_COMMENT = re.compile(rb'//[^\n]*|/\*.*?\*/|("(?:\\.|[^"\\\n])*")', re.S)


def _code_only(text: bytes) -> bytes:
    """The source without comments and with runs of white space collapsed: what the compiler's output depends on (string literals
    are kept as they are).  A reworded comment -- in particular in the public header, which every workload's fingerprint includes --
    must not make committed counters look stale."""
    text = _COMMENT.sub(lambda m: m.group(1) or b" ", text)
    return b" ".join(text.split())


def sources_fingerprint(files) -> str:
    """Hash of a SUBSET of the kernel sources (+ the public header and the compiler options), comments and white space apart: what
    one workload's kernels are compiled from.  bench.py stamps the committed PMC counters of a workload with it, so that a roofline
    fraction computed from counters of an older kernel is flagged (``counters_stale``) instead of silently wrong."""
    h = hashlib.sha256()
    for f in sorted(files) + ["../../include/pdegym.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(_code_only(fh.read()))
    h.update(" ".join(OPTS).encode())
    return h.hexdigest()[:16]


def library_stamp() -> str:
    """Fingerprint the shipped library was built from (the stamp file next to it), or "" if there is none."""
    try:
        return open(os.path.join(LIBDIR, "libpdegym_hip.stamp")).read().strip()
    except OSError:
        return ""


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link libpdegym_hip.so. Returns its path."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    stamp = os.path.join(LIBDIR, "libpdegym_hip.stamp")
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == fp:
        return LIB
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpdegym_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    with open(stamp, "w") as f:
        f.write(fp)
    return LIB


PROBE_SRC = os.path.join(ROOT, "tools", "hbm_probe.hip")
PROBE_LIB = os.path.join(LIBDIR, "libpdegym_probe.so")


def build_probe(force: bool = False) -> str:
    """The HBM yardstick (tools/hbm_probe.hip: float4 copy / read / fill) as its own small library: measurement tooling, not part
    of the product ABI, and kept out of csrc/ so that the product's kernel fingerprints do not depend on it."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(PROBE_SRC):
        if os.path.exists(PROBE_LIB):          # (an installed copy ships the built probe without tools/)
            return PROBE_LIB
        raise RuntimeError(f"{PROBE_SRC} is missing: the HBM probe is measurement tooling of the source tree, not of an installed package")
    if not force and os.path.exists(PROBE_LIB) and os.path.getmtime(PROBE_LIB) >= os.path.getmtime(PROBE_SRC):
        return PROBE_LIB
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpdegym_probe.so")
    os.makedirs(LIBDIR, exist_ok=True)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", PROBE_LIB, PROBE_SRC],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on hbm_probe.hip:\n" + r.stdout.decode())
    return PROBE_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_probe(force=True))
