#!/usr/bin/env python3
"""Differential fuzz of the 1D rollout kernels WITH THE POLICY INSIDE (pdegym_*_rollout, pdegym_rollout1d.policy) against the
two-launch path of the same HIP engines -- pdegym_mlp_forward on the observation (+ pre-drawn sensing noise), then one step
call -- for random networks (1 .. 4 layers, 1 .. 256 units, tanh / relu / identity, with and without bias, clamp, action noise),
random grids, sub-step counts, control / sensing combinations, batch sizes (incl. batches that are not a multiple of the 16
instances of a workgroup), rollout lengths and auto-reset pools.

A network with a layer of more than 64 units is evaluated by the workgroup's 16 waves together with pdegym_mlp_forward's own
MFMA reduction: EVERYTHING must agree bit for bit.  Narrow networks (one fma chain per neuron) agree to float32 rounding in
the first command; later steps are then compared by re-running the step calls on the commands the kernel issued (bitwise).

    python tests/fuzz_policy_rollout.py [seconds] [seed]
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
from pdecontrolgym_amd.policy import FusedMLP  # noqa: E402

DEV = "cuda"


def bits_equal(a, b):
    """Bit patterns (NaN == NaN of the same payload, -0.0 != +0.0)."""
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


def one_case(rng, idx=0):
    base = str(rng.choice(["parabolic", "transport"]))
    nx = int(rng.choice([3, 16, 31, 64, 65, 100, 128, 255, 256, 257, 300, 500, 512]))
    if base == "parabolic" and nx == 512:
        nx = 511            # rows of at most 513 nodes inside the policy kernels
    S = int(rng.choice([1, 2, 5, 10, 33, 100]))
    B, T = int(rng.choice([1, 3, 15, 16, 17, 37, 70])), int(rng.integers(1, 7))
    ep = int(rng.integers(1, 5))
    dx = 1.0 / nx
    dt = (0.25 * dx * dx if base == "parabolic" else 0.5 * dx) * float(rng.choice([1.0, 0.5]))
    nt_sub = max(ep * S - int(rng.integers(0, S)), 2)
    control = str(rng.choice(["Dirchilet", "Dirchilet", "Neumann"]))
    loc = str(rng.choice(["full", "full", "full", "collocated", "opposite"]))
    stype = None if loc != "opposite" else ("Neumann" if base == "parabolic" else str(rng.choice(["Neumann", "Dirchilet"])))
    kw = dict(T=nt_sub * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=control, sensing_loc=loc,
              sensing_type=stype, normalize=bool(rng.random() < 0.5), max_control_value=float(rng.choice([20, 1, 3])),
              limit_pde_state_size=bool(rng.random() < 0.7), max_state_value=float(rng.choice([1e10, 30.0, 1e3])))
    rk = int(rng.choice([N.REWARD_TUNED1D, N.REWARD_TUNED1D, N.REWARD_NORM_L1, N.REWARD_NORM_L2, N.REWARD_NORM_LINF]))
    nt1 = int(round(kw["T"] / dt))
    n = nx + (1 if base == "parabolic" else 0)
    x = np.linspace(0, 1, n)
    init = (rng.uniform(0.5, 3, (B, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (B, 1))))).astype(np.float32)
    beta = rng.uniform(-2, 2, (B, n)).astype(np.float32)
    P = int(rng.choice([B, 2 * B + 1]))
    pool = (rng.uniform(0.5, 3, (P, 1)) * np.ones((1, n))).astype(np.float32)

    def make():
        e = PDEBatch1D(base, reward=RewardSpec(rk, nt1, -1e3, 3e2), num_envs=B, device=DEV, **kw)
        e.reset(torch.tensor(init), torch.tensor(beta))
        e.enable_auto_reset(torch.tensor(pool), keep_final_obs=True)
        return e

    probe = make()
    od = probe.obs_dim
    wide = bool(rng.random() < 0.6)
    nl = int(rng.integers(1, 5))
    widths = [int(rng.choice([65, 80, 100, 128, 200, 255, 256]) if (wide and rng.random() < 0.7) else rng.choice([1, 3, 16, 17, 33, 48, 64]))
              for _ in range(nl - 1)] + [1]
    if wide and nl > 1 and max(widths) <= 64:
        widths[0] = int(rng.choice([65, 129, 256]))
    sizes = [od] + widths
    layers = []
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    for i in range(nl):
        lin = torch.nn.Linear(sizes[i], sizes[i + 1], bias=bool(rng.random() < 0.8))
        with torch.no_grad():
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * (1.5 / np.sqrt(sizes[i])))
            if lin.bias is not None:
                lin.bias.copy_(torch.randn(lin.bias.shape, generator=g) * 0.3)
        layers.append(lin)
        act = str(rng.choice(["tanh", "relu", "none"]))
        if act != "none":
            layers.append(torch.nn.Tanh() if act == "tanh" else torch.nn.ReLU())
    net = torch.nn.Sequential(*layers).to(DEV)
    clamp = (-1.0, 1.0) if rng.random() < 0.7 else None
    pol = FusedMLP(net, clamp=clamp)
    desc = (f"#{idx} {base} nx={nx} S={S} B={B} T={T} nt={nt1} {control} {loc}/{stype} reward={rk} net={sizes} clamp={clamp} "
            f"norm={kw['normalize']} limit={kw['limit_pde_state_size']}/{kw['max_state_value']}")
    if not probe.policy_fits_rollout(pol):
        return desc + " (policy does not fit: skipped)"
    use_an, use_sn = bool(rng.random() < 0.6), bool(rng.random() < 0.5)
    an = (torch.randn(T, B, generator=g) * 0.2).to(DEV) if use_an else None
    sn = (torch.randn(T, B, od, generator=g) * 0.05).to(DEV) if use_sn else None
    is_wide = max(widths) > 64

    def z(*shape, dt=torch.float32):
        return torch.zeros(*shape, dtype=dt, device=DEV)

    # (a) one launch
    ea = make()
    obs_a, act_a, rew_a, te_a, tr_a = z(T + 1, B, od), z(T, B), z(T, B), z(T, B, dt=torch.uint8), z(T, B, dt=torch.uint8)
    seen_a = z(T, B, od) if use_sn else None
    obs_a[0].copy_(ea.t["obs"].reshape(B, od))
    ea.rollout(obs_a, act_a, rew_a, te_a, tr_a, policy=pol, noise=an, obs_noise=sn, obs_seen=seen_a)
    # (b) policy launch + step launch per env-step; with a narrow network the kernel's own commands are replayed after the
    #     first comparison, so that a last-bit difference of a command does not grow into the trajectory
    eb = make()
    full = ea.sensing == N.SENSE_FULL
    cur = eb.t["obs"].reshape(B, od).clone()
    a_buf = z(B)
    for t in range(T):
        seen = cur if sn is None else cur + sn[t]
        pol.forward_into(seen.contiguous(), a_buf, noise=None if an is None else an[t].contiguous())
        if is_wide:
            assert bits_equal(a_buf, act_a[t]), desc + f" step {t}: commands differ (wide: must be bit-identical)"
        else:
            # one fma chain per neuron against the MFMA's order: the difference scales with the magnitude of the inputs
            scale = float(torch.nan_to_num(seen.abs(), nan=0.0, posinf=0.0).max().clamp(min=1.0))
            torch.testing.assert_close(a_buf, act_a[t], rtol=1e-4, atol=2e-5 * scale, equal_nan=True,
                                       msg=lambda m: desc + f" step {t} (max |input| {scale:.3g}): commands: " + m)
            a_buf.copy_(act_a[t])
        if use_sn:
            assert bits_equal(seen, seen_a[t]), desc + f" step {t}: obs_seen"
        o, r, te, tr = eb.step(a_buf)
        o = o.reshape(B, od)
        for name, x_, y_ in (("obs", o, obs_a[t + 1]), ("reward", r, rew_a[t]), ("terminated", te, te_a[t]), ("truncated", tr, tr_a[t])):
            xa, ya = x_.cpu().numpy(), y_.cpu().numpy()
            same = np.array_equal(xa.view(np.uint32), ya.view(np.uint32)) if xa.dtype == np.float32 else np.array_equal(xa, ya)
            assert same, desc + f" step {t}: {name}"
        cur = o.clone()
    assert torch.equal(ea.t["time_index"], eb.t["time_index"]), desc + " time_index"
    if not full:
        ua, ub = ea.t["u"].cpu().numpy(), eb.t["u"].cpu().numpy()
        if not np.array_equal(ua.view(np.uint32), ub.view(np.uint32)):
            bad = np.argwhere(ua.view(np.uint32) != ub.view(np.uint32))
            raise AssertionError(desc + f" u: {len(bad)} entries differ, first {bad[:4].tolist()}, a={ua[tuple(bad[0])]!r} b={ub[tuple(bad[0])]!r}; "
                                 f"te={te_a.cpu().numpy().tolist()} tr={tr_a.cpu().numpy().tolist()} t={ea.t['time_index'].cpu().numpy().tolist()}")
    return desc + (" [wide]" if is_wide else "")


def random_net(rng, g, n_in, n_out, wide):
    nl = int(rng.integers(1, 5))
    widths = [int(rng.choice([65, 80, 100, 128, 200, 255, 256]) if (wide and rng.random() < 0.7) else rng.choice([1, 3, 16, 17, 33, 48, 64]))
              for _ in range(nl - 1)] + [n_out]
    if wide and nl > 1 and max(widths) <= 64:
        widths[0] = int(rng.choice([65, 129, 256]))
    sizes = [n_in] + widths
    layers = []
    for i in range(nl):
        lin = torch.nn.Linear(sizes[i], sizes[i + 1], bias=bool(rng.random() < 0.8))
        with torch.no_grad():
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * (1.5 / np.sqrt(sizes[i])))
            if lin.bias is not None:
                lin.bias.copy_(torch.randn(lin.bias.shape, generator=g) * 0.3)
        layers.append(lin)
        act = str(rng.choice(["tanh", "relu", "none"]))
        if act != "none":
            layers.append(torch.nn.Tanh() if act == "tanh" else torch.nn.ReLU())
    return torch.nn.Sequential(*layers).to(DEV), sizes


def traffic_case(rng, idx=0):
    """pdegym_traffic_rollout with the policy inside (float64 observations rounded to float32 on their way in, one or two
    commands) against FusedMLP.forward_into + step calls; wide networks bit for bit."""
    sim = str(rng.choice(["inlet", "outlet", "both", "outlet-train"]))
    cf, B, T = int(rng.integers(1, 4)), int(rng.choice([1, 3, 16, 17, 40])), int(rng.integers(1, 8))
    X = float(rng.choice([500, 300, 630]))
    horizon = float(rng.choice([0.5, 2.0, 240.0]))
    rs = rng.choice([0.115, 0.12, 0.125], B)
    qclip = rs * (40 * (1 - rs / 0.16))
    A = 2 if sim == "both" else 1
    pool = rng.choice([0.115, 0.12, 0.125], int(rng.choice([1, B, 2 * B + 1])))
    auto = bool(rng.random() < 0.6)

    def make():
        e = TrafficBatch(horizon, 0.25, X, 10, sim, 40, 0.16, 60, True, cf, num_envs=B, device=DEV)
        e.set_action_bounds(qclip)
        e.reset(rs)
        if auto:
            e.enable_auto_reset(pool, keep_final_obs=True)
        return e

    ea = make()
    D = 2 * ea.M
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    wide = bool(rng.random() < 0.6)
    net, sizes = random_net(rng, g, D, A, wide)
    with torch.no_grad():
        last = [m for m in net if isinstance(m, torch.nn.Linear)][-1]
        if last.bias is not None:
            last.bias.add_(4.5)           # commands near the steady-state flux
    pol = FusedMLP(net, clamp=(3.0, 6.0) if rng.random() < 0.7 else None)
    desc = f"#{idx} traffic {sim} cf={cf} B={B} T={T} X={X} horizon={horizon} auto={auto} net={sizes}"
    if not ea.policy_fits_rollout(pol):
        return desc + " (policy does not fit: skipped)"
    is_wide = max(sizes[1:]) > 64
    an = (torch.randn(T, B, A, generator=g) * 0.05).to(DEV) if rng.random() < 0.6 else None
    f64 = torch.float64
    obs_a = torch.zeros(T + 1, B, D, dtype=f64, device=DEV)
    act_a = torch.zeros(T, B, A, dtype=f64, device=DEV)
    rew_a = torch.zeros(T, B, dtype=f64, device=DEV)
    dn_a, tr_a = torch.zeros(T, B, dtype=torch.uint8, device=DEV), torch.zeros(T, B, dtype=torch.uint8, device=DEV)
    obs_a[0].copy_(ea.t["obs"])
    ea.rollout(obs_a, act_a, rew_a, dn_a, tr_a, policy=pol, noise=an)
    eb = make()
    cur = eb.t["obs"].clone()
    a_buf = torch.zeros(B, A, dtype=f64, device=DEV)
    for t in range(T):
        pol.forward_into(cur, a_buf, noise=None if an is None else an[t].contiguous())
        if is_wide:
            assert torch.equal(a_buf.view(torch.int64), act_a[t].view(torch.int64)), desc + f" step {t}: commands differ (wide: must be bit-identical)"
        else:
            torch.testing.assert_close(a_buf, act_a[t], rtol=1e-4, atol=2e-5, msg=lambda m: desc + f" step {t}: commands: " + m)
            a_buf.copy_(act_a[t])
        o, r, d, c = eb.step(a_buf)
        for name, x_, y_ in (("obs", o, obs_a[t + 1]), ("reward", r, rew_a[t]), ("done", d, dn_a[t]), ("truncated", c, tr_a[t])):
            same = torch.equal(x_.contiguous().view(torch.uint8) if x_.dtype != torch.uint8 else x_,
                               y_.contiguous().view(torch.uint8) if y_.dtype != torch.uint8 else y_)
            assert same, desc + f" step {t}: {name}"
        cur = o.clone()
    for k in ("r", "y", "time", "rs"):
        assert torch.equal(ea.t[k].view(torch.int64), eb.t[k].view(torch.int64)), desc + " " + k
    return desc + (" [wide]" if is_wide else "")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, k, wide, skipped = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        d = traffic_case(rng, k) if k % 5 == 4 else one_case(rng, k)
        k += 1
        wide += d.endswith("[wide]")
        skipped += d.endswith(")")
        if k % 200 == 0:
            print(f"{k} cases ({wide} wide, {skipped} skipped), {time.time() - t0:.0f} s; last: {d}", flush=True)
    print(f"fuzz_policy_rollout: {k} cases ({wide} wide, {skipped} skipped), 0 mismatches in {time.time() - t0:.0f} s (seed {seed})")


if __name__ == "__main__":
    main()
