#!/usr/bin/env python3
"""Time the reference's shipped NavierStokes2D example configuration (NS2Dppo.py: 21x21, K=2000 Jacobi sweeps, float64)
with the column-per-lane kernel (ns_col_step) and with the workgroup-per-instance kernel:  python tools/bench_ns_example.py
(PDEGYM_NS_COL_MIN_BATCH=0 in the environment forces the column kernel at every batch size)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd.batch2d import NSBatch2D  # noqa: E402

BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"],
      "right": ["Dirchilet", "Dirchilet"]}


def _dbg(key, value):
    """Test-only kernel dispatch override (pdegym_debug_set, include/pdegym.h); takes ints or the "0"/"1" strings looped over."""
    from pdecontrolgym_amd import _native as N
    N.load().pdegym_debug_set(getattr(N, key), int(value))


def run(B, no_col, steps=20, n=21, K=2000, dtype=torch.float64):
    _dbg("DEBUG_NS_NO_COL", "1" if no_col else "0")
    nt = 200
    env = NSBatch2D(T=0.2, dt=1e-3, X=1, dx=1 / (n - 1), Y=1, dy=1 / (n - 1), boundary_condition=BC, U_ref=np.zeros((nt, n, n, 2)),
                    action_ref=2 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K, num_envs=B, device="cuda", dtype=dtype)
    z = np.zeros((B, n, n))
    env.reset(z, z, z)
    a = torch.full((B, 1), 3.0, dtype=dtype, device="cuda")
    for _ in range(3):
        env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(a)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    return el


if __name__ == "__main__":
    if len(sys.argv) > 1:       # other grid sizes: python tools/bench_ns_example.py 11 16 26 31   (K = 200, B = 8192)
        for n in map(int, sys.argv[1:]):
            for dtype in (torch.float64, torch.float32):
                for B in (512, 8192):
                    a, b = run(B, False, n=n, K=200, dtype=dtype), run(B, True, n=n, K=200, dtype=dtype)
                    print(f"{n}x{n} K=200 {str(dtype)[6:]} B={B}: column kernel {a*1e3:.3f} ms/step | workgroup kernel {b*1e3:.3f} ms/step")
        sys.exit(0)
    for dtype in (torch.float64, torch.float32):
        for B in (1, 1024, 3072, 8192, 32768):
            a, b = run(B, False, dtype=dtype), run(B, True, dtype=dtype)
            print(f"21x21 K=2000 {str(dtype)[6:]} B={B}: column kernel (B >= PDEGYM_NS_COL_MIN_BATCH) {a*1e3:.3f} ms/step ({B/a:.0f} env-steps/s)"
                  f" | workgroup kernel {b*1e3:.3f} ms/step ({B/b:.0f} env-steps/s)")
