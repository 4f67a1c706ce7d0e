#!/usr/bin/env python3
"""The HBM yardstick of this box: hand-written float4 copy / read / fill kernels (tools/hbm_probe.hip), 64 MiB ... 4 GiB, plain
and non-temporal, a sweep over the grid size -- the figure bench.py prints as `hbm_copy_measured_GBps` beside the 8 TB/s
specification (replaces the torch copy_ figure of tools/hbm_copy_bw.py, kept for comparison).

  python3 tools/hbm_probe.py [--json out.json]
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KINDS = {"copy": 0, "read": 1, "fill": 2}
_lib = None


def lib():
    global _lib
    if _lib is None:
        from pdecontrolgym_amd import build
        path = build.PROBE_LIB
        if not os.path.exists(path):
            path = build.build_probe()
        _lib = C.CDLL(path)
        _lib.pdegym_probe_hbm.restype = C.c_int
        _lib.pdegym_probe_hbm.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    return _lib


def run(kind, dst, src, nbytes, workgroups, nontemporal, reps=10):
    """Average seconds of one launch over `reps` back-to-back launches (HIP events on the current stream)."""
    import torch
    h = lib()
    stream = torch.cuda.current_stream().cuda_stream
    d = dst.data_ptr() if dst is not None else None
    s = src.data_ptr() if src is not None else None

    def launch():
        if h.pdegym_probe_hbm(KINDS[kind], d, s, nbytes, workgroups, int(nontemporal), stream) != 0:
            raise RuntimeError(f"pdegym_probe_hbm({kind}) failed")
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def measure(device="cuda", sizes_mib=(64, 512, 4096), grids=(2048, 4096, 8192, 16384, 65536), quick=False):
    """Best rate per kind and size over the grid sizes and the plain / non-temporal variants.  Returns
    {"copy_GBps": best read+write rate of the largest size, "read_GBps", "fill_GBps", "table": [...]}"""
    import torch
    if quick:
        sizes_mib, grids = (1024,), (4096, 16384)
    table = []
    best = {}
    for mib in sizes_mib:
        nbytes = mib << 20
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
        dst = torch.empty_like(src)
        acc = torch.zeros(1, dtype=torch.float32, device=device)
        for kind in ("copy", "read", "fill"):
            for nt in (False, True):
                for g in grids:
                    d = acc if kind == "read" else dst
                    t = run(kind, d, None if kind == "fill" else src, nbytes, g, nt, reps=5 if mib >= 2048 else 20)
                    moved = nbytes * (2 if kind == "copy" else 1)
                    row = {"kind": kind, "MiB": mib, "workgroups": g, "nontemporal": nt, "us": t * 1e6, "GBps": moved / t / 1e9}
                    table.append(row)
                    k = (kind, mib)
                    if k not in best or row["GBps"] > best[k]["GBps"]:
                        best[k] = row
        # the copy really copied (checked once per size, outside the timed launches; the fill runs above left dst = 1.0)
        lib().pdegym_probe_hbm(0, dst.data_ptr(), src.data_ptr(), nbytes, 4096, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(dst[:1 << 20], src[:1 << 20]) and torch.equal(dst[-(1 << 20):], src[-(1 << 20):])
        del src, dst
    big = max(sizes_mib)
    return {"copy_GBps": best[("copy", big)]["GBps"], "read_GBps": best[("read", big)]["GBps"], "fill_GBps": best[("fill", big)]["GBps"],
            "at_MiB": big, "best": [best[k] for k in sorted(best)], "table": table,
            "kernel": "tools/hbm_probe.hip: 16 B per lane and access, 4 independent loads in flight per lane, grid-stride"}


def torch_copy_GBps(device="cuda", mib=4096):
    import torch
    n = (mib << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3 / 10) / 1e9


if __name__ == "__main__":
    import torch
    res = measure()
    res["torch_copy_GBps"] = torch_copy_GBps()
    props = torch.cuda.get_device_properties(0)
    res["device"] = {"name": props.name, "CUs": props.multi_processor_count, "mem_GiB": props.total_memory // 2 ** 30}
    for r in res["best"]:
        print(f"{r['kind']:5s} {r['MiB']:5d} MiB: {r['GBps']:7.0f} GB/s  ({r['us']:.1f} us, {r['workgroups']} workgroups, nontemporal={r['nontemporal']})")
    print(f"torch copy_ 4 GiB: {res['torch_copy_GBps']:.0f} GB/s")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(res, fh, indent=1)
