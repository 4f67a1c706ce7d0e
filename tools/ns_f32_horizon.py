#!/usr/bin/env python3
"""Drift of the float32 NS2D throughput kernels against the float64 (reference-precision, bit-exact) kernels over an episode:
same initial state, same actions, N steps; prints max|u32 - u64| / max|u64| (and the same for p) at checkpoints.
The numbers back the float32 horizon stated in DESIGN.md and tested in tests/test_gpu_ns2d.py.
    python tools/ns_f32_horizon.py [n] [steps] [K]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd.batch2d import NSBatch2D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
B = 8
BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"],
      "right": ["Dirchilet", "Dirchilet"]}
dx = 1.0 / (n - 1)
dt = 0.2 * 0.5 * dx * dx / 0.1
nt = steps + 2
rng = np.random.default_rng(0)
xs = np.linspace(0, 1, n)
Xg, Yg = np.meshgrid(xs, xs)
u0 = np.stack([np.sin(2 * np.pi * Xg * rng.uniform(0.5, 2)) * np.cos(np.pi * Yg) * rng.uniform(0.5, 2) + rng.uniform(-1, 1) for _ in range(B)])
v0 = np.stack([np.cos(np.pi * Xg) * np.sin(2 * np.pi * Yg * rng.uniform(0.5, 2)) * rng.uniform(0.5, 2) + rng.uniform(-1, 1) for _ in range(B)])
u0[B // 2:] = rng.uniform(-5, 5, (B - B // 2, 1, 1))          # the bench's constant initial fields (NS2Dppo.py:14-18)
v0[B // 2:] = rng.uniform(-5, 5, (B - B // 2, 1, 1))
p0 = np.zeros((B, n, n))
kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=BC, U_ref=np.zeros((nt, n, n, 2)), action_ref=2.0 * np.ones(nt),
          gamma=0.1, maximum_pressure_iteration=K)
e64 = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float64, **kw)
e32 = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float32, **kw)
u32, v32 = u0.astype(np.float32), v0.astype(np.float32)
e64.reset(u32.astype(np.float64), v32.astype(np.float64), p0)
e32.reset(u32, v32, p0.astype(np.float32))
out = []
for t in range(1, steps + 1):
    a = rng.uniform(2, 4, B).astype(np.float32)
    o64, r64, _ = e64.step(a.astype(np.float64))
    o32, r32, _ = e32.step(a)
    if t in (1, 2, 5, 10, 20, 50, 100, 200, 400, 800, 1600) or t == steps:
        d = (o32.double() - o64).abs().amax(dim=(1, 2, 3))
        s = o64.abs().amax(dim=(1, 2, 3))
        dp = (e32.p.double() - e64.p).abs().amax(dim=(1, 2))
        sp = e64.p.abs().amax(dim=(1, 2))
        rr = ((r32.double() - r64).abs() / r64.abs().clamp_min(1e-30))
        rec = {"step": t, "u_rel_max": float((d / s).max()), "u_rel_smooth_ic": float((d / s)[: B // 2].max()), "u_rel_const_ic": float((d / s)[B // 2:].max()),
               "p_rel_max": float((dp / sp).max()), "reward_rel_max": float(rr.max()), "umax": float(s.max())}
        out.append(rec)
        print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/ns_f32_horizon_{n}.json", "w") as fh:
    json.dump({"n": n, "K": K, "B": B, "records": out}, fh, indent=1)
