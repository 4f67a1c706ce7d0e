#!/usr/bin/env python3
"""Quick tour of the drop-in API on an MI355X (every call below is the reference's own call shape).

    python examples/quickstart.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_control_gym  # noqa: E402  (registers the PDEControlGym-* ids)
from pde_control_gym.src import BrainTumorReward, NSReward, TunedReward1D  # noqa: E402

# ---- 1. one environment, exactly like examples/transportPDE/transport1Dppo.py:59-77 of the reference -------------------
T, dt, dx, X = 1, 1e-4, 1e-2, 1
beta = lambda nx: (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, nx)))).astype(np.float32)      # noqa: E731
params = {"T": T, "dt": dt, "X": X, "dx": dx, "reward_class": TunedReward1D(int(round(T / dt)), -1e3, 3e2), "normalize": True,
          "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": lambda s: s,
          "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20,
          "reset_init_condition_func": lambda nx: np.ones(nx) * 5.0, "reset_recirculation_func": beta, "control_sample_rate": 0.1}
env = pde_control_gym.make("PDEControlGym-TransportPDE1D", **params)
obs, info = env.reset()
total = 0.0
for k in range(10):
    obs, reward, terminated, truncated, info = env.step(np.array([0.1], dtype=np.float32))
    total += reward
print(f"TransportPDE1D: 10 env-steps (1000 PDE sub-steps each in one kernel launch), return {total:.3f}, terminated={terminated}")

# ---- 2. 4096 environments behind one VecEnv-style object, device tensors in and out ------------------------------------
venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=4096, **params)
venv.reset_tensor()
venv.enable_fused_auto_reset()
actions = torch.zeros(4096, device="cuda")
for k in range(10):
    obs_t, rew_t, term_t, trunc_t = venv.step_tensor(actions)
print(f"PDEVecEnv: obs {tuple(obs_t.shape)} on {obs_t.device}, mean reward {rew_t.mean().item():.4f}")

# ---- 3. policy forward + env step captured in one hipGraph ------------------------------------------------------------
policy = torch.nn.Sequential(torch.nn.Linear(100, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1), torch.nn.Tanh()).cuda()
rollout = pde_control_gym.DeviceRollout(venv, policy, n_steps=8).run()
print(f"DeviceRollout: buffers obs {tuple(rollout.obs.shape)}, rewards {tuple(rollout.rewards.shape)}")
# the same network as ONE launch per step (forward pass + action clamp + store), reading the module's own parameters
fused = pde_control_gym.FusedMLP(policy)
rollout = pde_control_gym.DeviceRollout(venv, fused, n_steps=8).run()
print(f"DeviceRollout + FusedMLP: actions in [{rollout.actions.min().item():.3f}, {rollout.actions.max().item():.3f}]")

# ---- 3b. output feedback: Neumann actuation, the collocated scalar measurement, sensing noise -- still ONE kernel per rollout -------
# (hyperbolic.py:66-124: control_type / sensing_loc pick one of the reference's ten variants; the policy's input is the one sensed value)
ofb = dict(params, control_type="Neumann", sensing_loc="collocated", sensing_noise_func=None)
venv2 = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=1024, **ofb)
venv2.reset_tensor()
venv2.enable_fused_auto_reset()
small = pde_control_gym.FusedMLP(torch.nn.Sequential(torch.nn.Linear(1, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1), torch.nn.Tanh()).cuda())
ro2 = pde_control_gym.DeviceRollout(venv2, small, n_steps=8, sensing_noise=True)
ro2.sensing_noise.normal_().mul_(0.01)             # drawn ahead by the caller; the policy reads obs + noise, obs_seen records it
ro2.run()
print(f"output-feedback rollout: one launch = {ro2.one_launch}, obs {tuple(ro2.obs.shape)}, max |obs_seen - obs| "
      f"{(ro2.obs_seen - ro2.obs).abs().max().item():.3f}")
# a checkpoint of the whole batch (device state as torch tensors; loads copy in place)
sd = venv2.state_dict()
before = venv2.step_tensor(torch.zeros(1024, device="cuda"))[0].clone()
venv2.load_state_dict(sd)
assert torch.equal(venv2.step_tensor(torch.zeros(1024, device="cuda"))[0], before)
print("checkpoint: state_dict() -> step -> load_state_dict() -> step reproduces the step bit for bit")

# ---- 4. Navier-Stokes, the parameter dictionary of examples/NavierStokes/NS2Dppo.py:36-50 ---------------------------------
bc = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"],
      "right": ["Dirchilet", "Dirchilet"]}
ns = pde_control_gym.make("PDEControlGym-NavierStokes2D", T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05, action_dim=1,
                          reward_class=NSReward(0.1), normalize=False, boundary_condition=bc, U_ref=np.zeros((200, 21, 21, 2)),
                          action_ref=2.0 * np.ones(1000), reset_init_condition_func=lambda Xg: (np.zeros_like(Xg),) * 3)
obs, _ = ns.reset()
obs, r, te, tr, _ = ns.step(3.0)
print(f"NavierStokes2D 21x21 float64 (2000 Jacobi sweeps per step): obs {obs.shape} {obs.dtype}, reward {r:.5f}")

# ---- 5. a cohort of brain-tumour patients under the batched TherapyWrapper -----------------------------------------------
def tumor_ic(X, nx):
    xs = np.linspace(0, X, nx)
    return 0.8 * 1e5 * np.exp(-0.25 * xs ** 2)


tvec = pde_control_gym.make_vec("PDEControlGym-BrainTumor1D", num_envs=1024, weekends=True, T=600, X=200, dt=1, dx=1,
                                reward_class=BrainTumorReward(), reset_init_condition_func=tumor_ic, total_dosage=61.2)
tb = tvec.benchmark()
tvec.reset_tensor()
dose = torch.full((1024,), 2.0 / 61.2, dtype=torch.float64, device="cuda")
finished = 0
for day in range(60):
    _, rew, term, trunc = tvec.step_tensor(dose)
    finished += int((term | trunc).sum())
print(f"TumorVecEnv: untreated survival {tb[0].item():.0f} days; 60 treatment steps for 1024 patients, {finished} episodes finished")
print("quickstart ok")
