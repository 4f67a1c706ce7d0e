#!/usr/bin/env python3
"""Throughput of the batched TherapyWrapper (TumorVecEnv): treatment steps/s seen by an agent and simulated patient-days/s
(growth and post-therapy stretches run inside the kernel).  python tools/bench_tumor_vec.py [B] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_control_gym  # noqa: E402
from pde_control_gym.src import BrainTumorReward  # noqa: E402


def ic(X, nx):
    xs = np.linspace(0, X, nx)
    return 0.8 * 1e5 * np.exp(-0.25 * (xs ** 2))


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kw = dict(T=600, X=200, dt=1, dx=1, reward_class=BrainTumorReward(), reset_init_condition_func=ic, total_dosage=61.2)
venv = pde_control_gym.make_vec("PDEControlGym-BrainTumor1D", num_envs=B, weekends=True, **kw)
t0 = time.perf_counter()
tb = venv.benchmark()
torch.cuda.synchronize()
print(f"benchmark(): {B} open-loop episodes ({int(tb.sum().item())} patient-days) in {time.perf_counter() - t0:.3f} s; survival {tb.min().item():.0f}..{tb.max().item():.0f} days")
venv.reset_tensor()
g = torch.Generator(device="cpu").manual_seed(0)
acts = (torch.rand(steps, B, generator=g, dtype=torch.float64) * 0.06).cuda()
for k in range(5):
    venv.step_tensor(acts[k])
torch.cuda.synchronize()
days0 = venv.core.t["time_index"].sum().item()
episodes, t0 = 0, time.perf_counter()
for k in range(steps):
    _, _, term, trunc = venv.step_tensor(acts[k])
    episodes += (term | trunc).sum()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"B={B}: {steps} VecEnv steps in {el:.3f} s = {B * steps / el:.4g} treatment-steps/s; {int(episodes)} episodes finished "
      f"(each one = ~213 growth + therapy + ~190 post-therapy days simulated in-kernel)")

# ---- the same loop with an MLP policy, captured once into a hipGraph and replayed (DeviceRollout)
torch.manual_seed(0)
pol = torch.nn.Sequential(torch.nn.Linear(venv.core.nx, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1), torch.nn.Sigmoid()).double().cuda()
policy = lambda o: pol(o / 1e5) * 0.06          # noqa: E731
T = 50
for use_graph in (False, True):
    venv.reset_tensor()
    ro = pde_control_gym.DeviceRollout(venv, policy, n_steps=T, use_graph=use_graph, action_low=0.0, action_high=1.0)
    ro.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        ro.run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"DeviceRollout graph={use_graph}: {B * T * reps / el:.4g} treatment-steps/s incl. the MLP policy ({el / (T * reps) * 1e6:.0f} us per VecEnv step)")
