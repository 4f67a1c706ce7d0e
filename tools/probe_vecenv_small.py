"""PDEVecEnv.step (the SB3-facing NumPy face) at small batches: microseconds per step() call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pde_control_gym
from pde_control_gym.src import TunedReward1D

def run(B, S, steps=400):
    nx = 100
    dt = 1e-4
    beta = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, nx)))).astype(np.float32)
    rng = np.random.default_rng(0)
    p = dict(T=5, dt=dt, X=1, dx=1e-2, reward_class=TunedReward1D(50000, -1e3, 3e2), normalize=True, sensing_loc="full",
             control_type="Dirchilet", sensing_type=None, sensing_noise_func=None, limit_pde_state_size=True, max_state_value=1e10,
             max_control_value=20, control_sample_rate=S * dt,
             batched_reset_func=lambda idx, nx_: (rng.uniform(1, 10, (len(idx), 1)).astype(np.float32) * np.ones((1, nx_), np.float32),
                                                  np.tile(beta, (len(idx), 1))))
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **p)
    venv.reset()
    acts = np.random.default_rng(1).uniform(-1, 1, (64, B, 1)).astype(np.float32)
    for k in range(20):
        venv.step(acts[k % 64])
    ts = []
    for k in range(steps):
        t0 = time.perf_counter()
        venv.step(acts[k % 64])
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6

for S in (1, 100):
    for B in (1, 8, 64, 512):
        print(f"S={S:4d} B={B:4d}: {run(B, S):7.1f} us per PDEVecEnv.step()", flush=True)
