#!/usr/bin/env python3
"""Developer probe: NS2D step time by grid size / dtype / Jacobi path (PDEGYM_NS_NO_LDS_JACOBI=1 forces the global-memory loop)."""
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


def run(n, K, B, dt_, no_lds):
    _dbg("DEBUG_NS_NO_LDS_JACOBI", "1" if no_lds else "0")
    dx = 1 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = 200
    env = NSBatch2D(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=BC, U_ref=np.zeros((nt, n, n, 2)),
                    action_ref=2 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K, num_envs=B, device="cuda", dtype=dt_)
    z = np.zeros((B, n, n))
    env.reset(z, z, z)
    a = torch.full((B, 1), 3.0, dtype=dt_, device="cuda")
    for _ in range(3):
        env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        env.step(a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20


if __name__ == "__main__":
    for n, K, B in ((21, 50, 512), (32, 50, 512), (45, 50, 512), (64, 50, 512), (64, 50, 64), (100, 50, 512), (128, 50, 512)):
        for dt_ in (torch.float64, torch.float32):
            if dt_ == torch.float32 and n in (64, 128):
                _dbg("DEBUG_NS_GENERIC", "1")
            a, b = run(n, K, B, dt_, False), run(n, K, B, dt_, True)
            _dbg("DEBUG_NS_GENERIC", "0")
            print(f"n={n} K={K} B={B} {str(dt_)[6:]} (generic kernel): LDS-path {a*1e3:.3f} ms | global-path {b*1e3:.3f} ms per step")
