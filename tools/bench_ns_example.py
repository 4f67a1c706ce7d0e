#!/usr/bin/env python3
"""Time the reference's shipped NavierStokes2D example configuration (NS2Dppo.py: 21x21, K=2000 Jacobi sweeps, float64)
with the LDS-resident Jacobi and with the global-memory loop:  python tools/bench_ns_example.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd.batch2d import NSBatch2D  # noqa: E402

BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"],
      "right": ["Dirchilet", "Dirchilet"]}


def run(B, no_lds, steps=50, n=21, K=2000, dtype=torch.float64):
    os.environ["PDEGYM_NS_NO_LDS_JACOBI"] = "1" if no_lds else "0"
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
    for B in (1, 64, 1024, 8192):
        a, b = run(B, False), run(B, True)
        print(f"21x21 K=2000 f64 B={B}: LDS {a*1e3:.3f} ms/step ({B/a:.0f} env-steps/s) | global {b*1e3:.3f} ms/step ({B/b:.0f} env-steps/s)")
