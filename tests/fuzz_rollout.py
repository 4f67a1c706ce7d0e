#!/usr/bin/env python3
"""Differential fuzz of the rollout kernels (pdegym_*_rollout, pdegym_traffic_rollout) against T step calls of the same HIP
engines (which tests/fuzz_1d.py / fuzz_more.py pin to the oracle): random grids, sub-step counts, episode lengths, batch
sizes, rollout lengths, reward kinds, normalisation, truncation thresholds, auto-reset pools (initial conditions and beta),
zero / tiny / large states, every control / sensing combination of the reference's table (round 4: Neumann actuation, the four
scalar sensing modes); all four traffic simulation types.  Everything must agree bit for bit.

    python tests/fuzz_rollout.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N  # noqa: E402
from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec  # noqa: E402
from pdecontrolgym_amd.batch_traffic import TrafficBatch  # noqa: E402

DEV = "cuda"


def case_1d(rng):
    kind = str(rng.choice(["parabolic", "transport", "burgers"]))
    base = "parabolic" if kind == "parabolic" else "transport"
    nx = int(rng.choice([3, 4, 31, 63, 64, 65, 100, 128, 129, 255, 256, 257, 300, 512, 513, 700, 1024, 1025, 1500, 2040])) if rng.random() < 0.6 \
        else int(rng.integers(3, 2040))
    S = int(rng.choice([1, 2, 3, 7, 10, 33, 100, 101, 128]))
    B, T = int(rng.choice([1, 2, 3, 5, 9, 17, 33])), int(rng.integers(1, 9))
    ep = int(rng.integers(1, 6))                       # env-steps per episode
    extra = int(rng.integers(0, S))
    dx = 1.0 / nx
    dt = (0.25 * dx * dx if base == "parabolic" else 0.5 * dx) * float(rng.choice([1.0, 0.5, 0.9]))
    nt_sub = max(ep * S - extra, 2)
    control = str(rng.choice(["Dirchilet", "Dirchilet", "Neumann"]))
    loc = str(rng.choice(["full", "full", "collocated", "opposite"]))
    stype = None if loc != "opposite" else ("Neumann" if base == "parabolic" else str(rng.choice(["Neumann", "Dirchilet"])))
    kw = dict(T=nt_sub * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=control, sensing_loc=loc,
              sensing_type=stype, normalize=bool(rng.random() < 0.5), max_control_value=float(rng.choice([20, 1, 3])),
              limit_pde_state_size=bool(rng.random() < 0.7), max_state_value=float(rng.choice([1e10, 30.0, 5.0, 1e3])))
    if kind == "burgers":
        kw["flux"] = "burgers"
    rk = int(rng.choice([N.REWARD_TUNED1D, N.REWARD_TUNED1D, N.REWARD_NORM_L1, N.REWARD_NORM_L2, N.REWARD_NORM_LINF, N.REWARD_NONE]))
    nt1 = int(round(kw["T"] / dt))
    n = nx + (1 if base == "parabolic" else 0)
    x = np.linspace(0, 1, n)
    style = str(rng.choice(["smooth", "const", "zeros", "tiny", "big"]))
    scale = {"smooth": 1.0, "const": 1.0, "zeros": 0.0, "tiny": 1e-38, "big": 1e30}[style]

    def rows(k):
        if style == "smooth":
            return (rng.uniform(0.5, 5, (k, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (k, 1))))).astype(np.float32)
        return (scale * rng.uniform(-3, 10, (k, 1)) * np.ones((1, n))).astype(np.float32)

    shared_beta = bool(rng.random() < 0.3)
    beta = rng.uniform(-2, 2, (n,) if shared_beta else (B, n)).astype(np.float32)
    P = int(rng.choice([B, 2 * B, 3 * B + 1]))
    auto = bool(rng.random() < 0.8)
    bpool = auto and not shared_beta and bool(rng.random() < 0.5)
    init, pool_i = rows(B), rows(P)
    pool_b = rng.uniform(-2, 2, (P, n)).astype(np.float32)
    acts = torch.tensor(rng.uniform(-1.5, 1.5, (T, B)).astype(np.float32), device=DEV)
    outs = []
    for mode in ("steps", "rollout"):
        e = PDEBatch1D(base, reward=RewardSpec(rk, nt1, -1e3, 3e2) if rk != N.REWARD_NONE else None, num_envs=B, device=DEV, **kw)
        if not e.can_rollout():
            return None
        e.reset(torch.tensor(init), torch.tensor(beta))
        if auto:
            e.enable_auto_reset(torch.tensor(pool_i), keep_final_obs=bool(rng.random() < 2), beta_pool=torch.tensor(pool_b) if bpool else None)
        obs = torch.zeros(T + 1, B, e.obs_dim, device=DEV)
        obs[0].copy_(e.t["obs"])
        rew = torch.zeros(T, B, device=DEV)
        te = torch.zeros(T, B, dtype=torch.uint8, device=DEV)
        tr = torch.zeros(T, B, dtype=torch.uint8, device=DEV)
        if mode == "steps":
            if e.state_in_obs:
                e.t["obs"] = obs[0]
                e.t["u"] = obs[0]
            for t in range(T):
                e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
        else:
            e.rollout(obs, acts, rew, te, tr)
        keep = [obs, rew, te, tr] + [e.t[k] for k in ("time_index", "bsum", "ring", "norm_now", "norm_back", "beta", "u") if torch.is_tensor(e.t.get(k))]
        keep += [e.t[k] for k in ("reset_count", "final_obs") if torch.is_tensor(e.t.get(k))]
        outs.append([k.clone() for k in keep])
    desc = f"{kind} {control}/{loc}/{stype} nx={nx} S={S} B={B} T={T} ep={ep} rk={rk} style={style} auto={auto} bpool={bpool} shared_beta={shared_beta}"
    for i, (a, b) in enumerate(zip(*outs)):
        if not torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b):
            # NaN rows compare unequal as floats; the byte views above already handle that -- a real mismatch
            raise AssertionError(f"1D rollout != steps (tensor {i}): {desc}")
    return desc


def case_traffic(rng):
    sim = str(rng.choice(["inlet", "outlet", "both", "outlet-train"]))
    cf, B, T = int(rng.integers(1, 5)), int(rng.choice([1, 3, 16, 17, 70])), int(rng.integers(1, 20))
    X = float(rng.choice([500, 300, 630]))
    horizon = float(rng.choice([0.5, 2.0, 240.0]))
    rs = rng.choice([0.115, 0.12, 0.125], B)
    qclip = rs * (40 * (1 - rs / 0.16))
    nact = 2 if sim == "both" else 1
    acts = torch.tensor(rng.uniform(0.5, 1.5, (T, B, nact)) * qclip[None, :, None], device=DEV)
    auto = bool(rng.random() < 0.6)
    pool = rng.choice([0.115, 0.12, 0.125], int(rng.choice([1, B, 2 * B + 1])))
    keep_final = bool(rng.random() < 0.7)
    outs = []
    for mode in ("steps", "rollout"):
        env = TrafficBatch(horizon, 0.25, X, 10, sim, 40, 0.16, 60, bool(rng.random() < 2), cf, num_envs=B, device=DEV)
        if not env.can_rollout():
            return None
        env.set_action_bounds(qclip)
        env.reset(rs)
        if auto:
            env.enable_auto_reset(pool, keep_final_obs=keep_final)
        D = 2 * env.M
        obs = torch.zeros(T + 1, B, D, dtype=torch.float64, device=DEV)
        obs[0].copy_(env.t["obs"])
        rew = torch.zeros(T, B, dtype=torch.float64, device=DEV)
        dn = torch.zeros(T, B, dtype=torch.uint8, device=DEV)
        tr = torch.zeros(T, B, dtype=torch.uint8, device=DEV)
        if mode == "steps":
            for t in range(T):
                o, r, d, c = env.step(acts[t])
                obs[t + 1].copy_(o), rew[t].copy_(r), dn[t].copy_(d), tr[t].copy_(c)
        else:
            env.rollout(obs, acts, rew, dn, tr)
        outs.append([k.clone() for k in (obs, rew, dn, tr, env.t["r"], env.t["y"], env.t["time"], env.t["rs"])]
                    + [env.t[k].clone() for k in ("reset_count", "final_obs") if env.t.get(k) is not None])
    desc = f"traffic {sim} cf={cf} B={B} T={T} X={X} horizon={horizon} auto={auto} pool={len(pool)}"
    for i, (a, b) in enumerate(zip(*outs)):
        if not torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b):
            raise AssertionError(f"traffic rollout != steps (tensor {i}): {desc}")
    return desc


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, done, skipped, last = time.time(), {"1d": 0, "traffic": 0}, 0, ""
    while time.time() - t0 < seconds:
        which = "traffic" if rng.random() < 0.3 else "1d"
        with np.errstate(all="ignore"):
            d = case_traffic(rng) if which == "traffic" else case_1d(rng)
        if d is None:
            skipped += 1
            continue
        done[which] += 1
        last = d
        if sum(done.values()) % 200 == 0:
            print(f"{done} ok ({skipped} skipped)  last: {last}", flush=True)
    print(f"FUZZ OK: {done}, {skipped} skipped, seed {seed}")


if __name__ == "__main__":
    main()
