#!/usr/bin/env python3
"""PPO on a batched PDEControlGym environment with NOTHING on the host inside a rollout.

The reference trains its RL controllers with SB3 on one environment per Python call
(examples/ReactionDiffusionPDE/..., examples/transportPDE/transport1Dppo.py:77-90: PPO("MlpPolicy", env)).  This is the same
algorithm in plain torch on the batched engine:

  * rollout  -- ``DeviceRollout``: the whole T-step rollout is ONE kernel launch (``pdegym_parabolic_rollout``: per env-step
                the actor mean is evaluated inside the kernel from weights held in LDS, exploration noise added, the command
                clamped and stored, then the env-step with fused auto-reset; layers of 65 .. 256 units -- ``hidden`` below --
                are evaluated in the same kernel by the 16 waves of a workgroup together on the matrix cores; anything else
                falls back to one ``pdegym_mlp_forward`` launch + one env-step launch per step inside a hipGraph);
  * update   -- ordinary torch autograd on the same ``torch.nn.Sequential`` the rollout evaluates (``FusedMLP`` picks the new
                weights up at the next ``run()``).

The plant is the reference's unstable reaction-diffusion benchmark u_t = u_xx + beta(x) u with boundary actuation
(ReactionDiffusionPDE1D, Dirichlet control at x = 1).  Uncontrolled, ||u|| grows; the printed mean ||u|| over an episode falls
as the policy learns to damp it.

    python examples/train_ppo_device.py [iterations] [num_envs] [hidden units per layer: 64]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_control_gym  # noqa: E402
from pde_control_gym import DeviceRollout, FusedMLP  # noqa: E402
from pde_control_gym.src import TunedReward1D  # noqa: E402


def make_env(B, nx=64, S=25, horizon=64):
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    beta = np.full(nx + 1, 12.0, np.float32)            # lambda = 12 > pi^2: the open loop is unstable
    rng = np.random.default_rng(0)

    def batched_reset(idx, n):
        amp = rng.uniform(1.0, 5.0, (len(idx), 1)).astype(np.float32)
        x = np.linspace(0, 1, n + 1, dtype=np.float32)[None, :]
        return amp * np.sin(np.pi * x).astype(np.float32) + amp * 0.2, np.tile(beta, (len(idx), 1))

    kw = {"T": horizon * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(horizon * S, -1e-2, 20.0),
          "normalize": True, "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
          "limit_pde_state_size": True, "max_state_value": 1e3, "max_control_value": 10.0, "control_sample_rate": S * dt,
          "batched_reset_func": batched_reset}
    return pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, **kw)


def mlp(sizes, out_tanh=False):
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(torch.nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2 or out_tanh:
            mods.append(torch.nn.Tanh())
    return torch.nn.Sequential(*mods)


def main(iterations=30, B=2048, T=64, quiet=False, hidden=64):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    venv = make_env(B, horizon=T)        # one rollout = one episode of every environment
    obs0 = venv.reset_tensor()
    venv.enable_fused_auto_reset()
    D = obs0.shape[1]
    # mean of a Gaussian policy: SB3's MlpPolicy shape (64-64), or e.g. hidden = 256 for net_arch [256, 256] -- layers of more
    # than 64 units are evaluated inside the same rollout kernel by the workgroup's 16 waves together (matrix cores)
    actor = mlp([D, hidden, hidden, 1]).to(dev)
    critic = mlp([D, hidden, hidden, 1]).to(dev)
    with torch.no_grad():
        actor[0].weight.mul_(0.1)                        # observations are O(10)
        critic[0].weight.mul_(0.1)
        actor[-1].weight.mul_(0.01)
    log_std = torch.nn.Parameter(torch.full((1,), -0.7, device=dev))
    opt = torch.optim.Adam(list(actor.parameters()) + list(critic.parameters()) + [log_std], lr=1e-3)
    rollout = DeviceRollout(venv, FusedMLP(actor), T, action_noise=True)
    gamma, lam, clip, epochs, nmb = 0.99, 0.95, 0.2, 6, 4
    history = []
    t_roll = t_upd = 0.0
    for it in range(iterations):
        # ---- rollout: one graph replay, no host work per step --------------------------------------------------------
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        std_old = log_std.detach().exp()
        rollout.action_noise.normal_().mul_(std_old)
        rollout.run()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - t0
        t0 = time.perf_counter()
        obs, rew = rollout.obs, rollout.rewards
        done = (rollout.terminated | rollout.truncated).float()
        with torch.no_grad():
            val = critic(obs.reshape(-1, D)).reshape(T + 1, B)
            adv = torch.zeros(T, B, device=dev)
            last = torch.zeros(B, device=dev)
            for t in reversed(range(T)):                  # GAE; an episode end cuts the bootstrap
                nonterm = 1.0 - done[t]
                delta = rew[t] + gamma * val[t + 1] * nonterm - val[t]
                last = delta + gamma * lam * nonterm * last
                adv[t] = last
            ret = adv + val[:T]
            mean_old = actor(obs[:T].reshape(-1, D)).reshape(T, B)
            a_raw = mean_old + rollout.action_noise       # the sample before the clamp: what the log-probabilities refer to
            logp_old = -0.5 * ((a_raw - mean_old) / std_old) ** 2 - log_std.detach()
        flat = lambda x: x.reshape(T * B, *x.shape[2:])   # noqa: E731
        o_f, a_f, lp_f, adv_f, ret_f = flat(obs[:T]), flat(a_raw), flat(logp_old), flat(adv), flat(ret)
        adv_f = (adv_f - adv_f.mean()) / (adv_f.std() + 1e-8)
        for _ in range(epochs):
            perm = torch.randperm(T * B, device=dev)
            for mb in perm.chunk(nmb):
                mean = actor(o_f[mb]).squeeze(-1)
                logp = -0.5 * ((a_f[mb] - mean) / log_std.exp()) ** 2 - log_std
                ratio = (logp - lp_f[mb]).exp()
                pg = -torch.min(ratio * adv_f[mb], ratio.clamp(1 - clip, 1 + clip) * adv_f[mb]).mean()
                vloss = 0.5 * (critic(o_f[mb]).squeeze(-1) - ret_f[mb]).pow(2).mean()
                loss = pg + 0.5 * vloss - 1e-3 * log_std.sum()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(list(actor.parameters()) + list(critic.parameters()), 0.5)
                opt.step()
        torch.cuda.synchronize()
        t_upd += time.perf_counter() - t0
        stats = {"iter": it, "episode_return": rew.sum(0).mean().item(), "mean_norm": obs[:T].norm(dim=2).mean().item(),
                 "truncated_frac": rollout.truncated.float().mean().item() * T, "std": log_std.exp().item()}
        history.append(stats)
        if not quiet and (it % 5 == 0 or it == iterations - 1):
            print(f"iter {it:3d}  episode return {stats['episode_return']:+8.3f}  mean ||u|| over the episode {stats['mean_norm']:7.3f}  "
                  f"episodes cut off per env {stats['truncated_frac']:.3f}  policy std {stats['std']:.3f}")
    if not quiet:
        print(f"{iterations} iterations x {T} steps x {B} envs: rollouts {t_roll:.2f} s ({iterations * T * B / t_roll:.3g} env-steps/s incl. policy), "
              f"updates {t_upd:.2f} s")
    return history


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 2048,
         hidden=int(sys.argv[3]) if len(sys.argv) > 3 else 64)
