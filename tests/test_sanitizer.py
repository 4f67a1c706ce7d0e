"""Host-only sanitizer job for the C-ABI shim.

GPU AddressSanitizer is not available on the pool, and the device code is covered by the parity suites; what a sanitizer CAN
check here is the HOST side of libpdegym_hip.so -- descriptor validation, kernel selection, launch-parameter arithmetic, the
thread-local error slot.  This test compiles the host half of every translation unit (``hipcc --cuda-host-only``: a few seconds)
with ``-fsanitize=address,undefined -fno-sanitize-recover``, links ``libpdegym_hip_asan.so`` into a scratch directory and runs
``tests/c/abi_validation.c`` -- 74 calls over every exported entry point with null / out-of-range / inconsistent arguments and
with well-formed descriptors on a machine without a device.  Any sanitizer report aborts the child; any call that does not
answer with a negative code and a message fails the test.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_host_side_of_the_c_abi_under_asan_and_ubsan(tmp_path):
    from pdecontrolgym_amd import build
    hipcc = shutil.which("hipcc")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + build.CSRC]
    objs, procs = [], []
    for s in build.SOURCES:
        o = str(tmp_path / s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC"] + SAN + inc + [
            "-c", os.path.join(build.CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, f"{s}:\n{out.decode()[-3000:]}"
    # a host-only object still refers to its (absent) device image, ``__hip_fatbin_<hash>``: give each an EMPTY offload bundle
    # (magic + zero entries) -- the runtime then finds no code object for any kernel and every launch fails cleanly
    nm = subprocess.run(["nm", "-u"] + objs, stdout=subprocess.PIPE, check=True).stdout.decode()
    names = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin_" in ln})
    assert len(names) >= len(build.SOURCES) - 1, names
    stub = tmp_path / "empty_fatbins.c"
    stub.write_text("".join(f'__attribute__((aligned(4096))) const char {n}[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n' for n in names))
    stub_o = str(tmp_path / "empty_fatbins.o")
    subprocess.run(["gcc", "-c", "-fPIC", str(stub), "-o", stub_o], check=True)
    objs.append(stub_o)
    lib = str(tmp_path / "libpdegym_hip_asan.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC"] + SAN + ["-o", lib] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    exe = str(tmp_path / "abi_validation")
    r = subprocess.run([hipcc, "-x", "c", "-std=c11", "-Wall", "-Werror", "-g"] + SAN + [os.path.join(ROOT, "tests", "c", "abi_validation.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + str(tmp_path), "-lpdegym_hip_asan",
                        "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")      # "no device" also on a GPU box: only the host paths are meant
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0 and "VALIDATION-OK" in out, out[-6000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-6000:]
    n = int(out.strip().splitlines()[-1].split()[1])
    assert n >= 70, out[-2000:]
