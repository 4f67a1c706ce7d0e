import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, bench
import pde_control_gym
from pde_control_gym.src import TunedReward1D
nx, S, B = 256, 100, 4096
dx = 1.0 / nx; dt = 0.25 * dx * dx
beta = (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)
rng = np.random.default_rng(0)
p = {"T": 1000 * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(1000 * S, -1e3, 3e2), "normalize": True,
     "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
     "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20, "control_sample_rate": S * dt,
     "batched_reset_func": lambda idx, nx_: (rng.uniform(1, 10, (len(idx), 1)).astype(np.float32) * np.ones((1, nx_ + 1), np.float32), np.tile(beta, (len(idx), 1)))}
venv = pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, device="cuda", **p)
venv.reset(); venv.enable_fused_auto_reset()
a = np.zeros((B, 1), np.float32)
for k in range(10): venv.step(a)
kept, ts = [], []
for k in range(40):
    t0 = time.perf_counter(); kept.append(venv.step(a)[0]); ts.append((time.perf_counter() - t0) * 1e6)
print([round(t) for t in ts])
