#!/usr/bin/env python3
"""NavierStokes2D 21 x 21 (the reference's shipped grid): T env-steps as T step launches replayed from one hipGraph against ONE
pdegym_ns2d_rollout launch, for several sweep counts K.  Prints microseconds per env-step of the whole batch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd.batch2d import NSBatch2D  # noqa: E402

BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}


def run(K, B, T, dtype):
    n, nt = 21, 100000
    dx = 1.0 / (n - 1)
    dt = 1e-3
    td = getattr(torch, dtype)
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=BC, U_ref=torch.zeros(8, n, n, 2, dtype=td, device="cuda"),
              action_ref=2.0 * torch.ones(8, dtype=td, device="cuda"), gamma=0.1, maximum_pressure_iteration=K)
    rng = np.random.default_rng(0)
    ic = [rng.uniform(-1, 1, (B, 1, 1)) * np.ones((1, n, n)) for _ in range(3)]
    acts = torch.as_tensor(rng.uniform(2, 4, (T, B, 1)), dtype=td, device="cuda")
    res = {}
    for mode in ("graph of step launches", "one rollout launch"):
        env = NSBatch2D(num_envs=B, device="cuda", dtype=td, **kw)
        env.reset(*ic)
        obs = torch.zeros(T + 1, B, n, n, 2, dtype=td, device="cuda")
        obs[0].copy_(env.t["obs"])
        rew = torch.zeros(T, B, dtype=td, device="cuda")
        te = torch.zeros(T, B, dtype=torch.uint8, device="cuda")

        def body():
            if mode.startswith("graph"):
                env.t["obs"] = obs[0]
                for t in range(T):
                    env.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t])
            else:
                env.rollout(obs, acts, rew, te)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            body()
            with torch.cuda.graph(g, stream=side):
                body()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        reps = max(3, int(0.2 / max(1e-6, T * 30e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / reps / T * 1e6
    a, b = res["graph of step launches"], res["one rollout launch"]
    print(f"21x21 {dtype} K={K} B={B} T={T}: {a:.1f} us per env-step as step launches | {b:.1f} us in one rollout launch ({a / b:.2f}x)", flush=True)


if __name__ == "__main__":
    for dtype in ("float64", "float32"):
        for K, B in ((10, 8192), (50, 8192), (50, 1024), (200, 8192), (2000, 3072)):
            run(K, B, 32 if K < 2000 else 8, dtype)
