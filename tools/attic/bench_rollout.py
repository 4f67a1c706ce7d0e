#!/usr/bin/env python3
"""Launch-bound regime (S = 1 sub-step per env-step): env-steps/s of a full on-device rollout (MLP policy forward +
fused env step with auto-reset) replayed from one hipGraph vs the same loop driven from Python."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pde_control_gym
from pde_control_gym import DeviceRollout
from pde_control_gym.src import TunedReward1D

B, T, S = 4096, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 1
nx = 256
dx = 1.0 / nx
dt = 0.25 * dx * dx
beta = (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)
p = {"T": 1000 * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(1000 * S, -1e3, 3e2), "normalize": True,
     "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
     "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20, "control_sample_rate": S * dt,
     "batched_reset_func": lambda idx, nx_: (np.random.default_rng(0).uniform(1, 10, (len(idx), 1)).astype(np.float32) * np.ones((1, nx_ + 1), np.float32),
                                             np.tile(beta, (len(idx), 1)))}
pol = torch.nn.Sequential(torch.nn.Linear(nx + 1, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                          torch.nn.Linear(64, 1), torch.nn.Tanh()).cuda()
res = {}
for graph in (False, True):
    venv = pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, **p)
    venv.reset_tensor()
    venv.enable_fused_auto_reset()
    ro = DeviceRollout(venv, pol, T, use_graph=graph)
    ro.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ro.run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    res["graph" if graph else "eager"] = {"env_steps_per_s": B * T / el, "us_per_step": el / T * 1e6}
print(json.dumps({"config": f"ReactionDiffusionPDE1D nx=256 B={B} S={S}, MLP 257-64-64-1 policy, T={T}", **res}))
