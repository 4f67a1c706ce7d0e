"""Host cost of one PDEBatch1D.step() call (argument marshalling + launch, no synchronisation inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pdecontrolgym_amd import _native as N
from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
dev = torch.device("cuda", 0)
for B in (8, 4096):
    nx, S = 256, 1
    dx = 1.0 / nx; dt = 0.25 * dx * dx
    e = PDEBatch1D("parabolic", 1000 * dt, dt, 1, dx, S * dt, normalize=True, limit_pde_state_size=True, reward=RewardSpec(N.REWARD_TUNED1D, 1000, -1e3, 3e2),
                   num_envs=B, device=dev)
    e.reset(torch.ones(B, nx + 1), torch.ones(nx + 1))
    a = torch.zeros(B, device=dev)
    for _ in range(50): e.step(a)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(300): e.step(a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"B={B}: {1e6 * (t1 - t0) / 300:.1f} us per step() call on the host (queue drained in {1e6 * (t2 - t1):.0f} us more)")
    bk = e.backend
    T = e.t
    t0 = time.perf_counter()
    for _ in range(300): bk._bufs1d(T)
    print(f"   _bufs1d alone: {1e6 * (time.perf_counter() - t0) / 300:.1f} us")
