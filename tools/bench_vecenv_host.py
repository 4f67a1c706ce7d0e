#!/usr/bin/env python3
"""The SB3-facing path: PDEVecEnv.step(numpy actions) -> numpy observations / rewards / dones / infos, host round trip included
(what PPO("MlpPolicy", venv).learn() drives).  python tools/bench_vecenv_host.py [B]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_control_gym  # noqa: E402
from pde_control_gym.src import TunedReward1D  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nx, S = 256, 100
dx = 1.0 / nx
dt = 0.25 * dx * dx
beta = (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)
p = {"T": 1000 * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(1000 * S, -1e3, 3e2), "normalize": True,
     "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
     "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20, "control_sample_rate": S * dt,
     "batched_reset_func": lambda idx, nx_: (np.random.default_rng(0).uniform(1, 10, (len(idx), 1)).astype(np.float32) * np.ones((1, nx_ + 1), np.float32),
                                             np.tile(beta, (len(idx), 1)))}
venv = pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, **p)
venv.reset()
venv.enable_fused_auto_reset()
acts = np.random.default_rng(1).uniform(-1, 1, (64, B, 1)).astype(np.float32)
for k in range(8):
    venv.step(acts[k])
t0 = time.perf_counter()
n = 56
for k in range(n):
    obs, rew, dones, infos = venv.step(acts[8 + k])
el = (time.perf_counter() - t0) / n
print(f"PDEVecEnv.step, B={B}, nx=256, S=100 (numpy in, numpy out): {el * 1e6:.0f} us per step = {B / el:.3g} env-steps/s "
      f"({obs.nbytes / 1e6:.1f} MB of observations to the host per step)")
