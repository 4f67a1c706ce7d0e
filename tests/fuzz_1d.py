#!/usr/bin/env python3
"""Differential fuzz of the 1D step kernels against the NumPy oracle (test infrastructure; run on a GPU box):
random grid sizes, sub-step counts, episode lengths (incl. nt <= 128 and episodes that end mid-step), control / sensing
modes, normalisation, truncation thresholds, shared or per-instance beta, zero / tiny / large states, history recording and
the fused auto-reset.  Rows, observations and flags must agree bit for bit, rewards to rtol 1e-6.

    python tests/fuzz_1d.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pde_oracle as po  # noqa: E402
from pdecontrolgym_amd import _native as N  # noqa: E402
from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec  # noqa: E402


def one_case(rng, idx):
    kind = rng.choice(["parabolic", "transport", "burgers"])
    flux = "burgers" if kind == "burgers" else "linear"
    base = "parabolic" if kind == "parabolic" else "transport"
    nx = int(rng.choice([3, 4, 5, 31, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025,
                         1500, 2047, 2048, 2049, 2500, 4096])) if rng.random() < 0.6 else int(rng.integers(3, 2600))
    S = int(rng.choice([1, 2, 3, 7, 10, 33, 100, 101, 128, 250]))
    nsteps = int(rng.integers(1, 7))
    extra = int(rng.integers(0, S))                      # the last step of the episode is cut short by `extra` sub-steps
    dx = 1.0 / nx
    dt = (0.25 * dx * dx if base == "parabolic" else 0.5 * dx) * float(rng.choice([1.0, 0.5, 0.9]))
    nt_sub = max(nsteps * S - extra, 1)
    ctrl = str(rng.choice(["Dirchilet", "Neumann"]))
    sloc = str(rng.choice(["full", "collocated", "opposite"]))
    stype = str(rng.choice(["Dirchilet", "Neumann"]))
    if base == "parabolic" and sloc == "opposite" and stype == "Dirchilet":
        stype = "Neumann"
    normalize = bool(rng.random() < 0.5)
    limit = bool(rng.random() < 0.7)
    max_state = float(rng.choice([1e10, 30.0, 5.0, 1e3]))
    kw = dict(T=nt_sub * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=ctrl, sensing_loc=sloc,
              sensing_type=None if sloc == "full" else stype, normalize=normalize, max_control_value=float(rng.choice([20, 1, 3])),
              limit_pde_state_size=limit, max_state_value=max_state)
    B = int(rng.choice([1, 2, 3, 4, 5, 9]))
    n = nx + (1 if base == "parabolic" else 0)
    nt = int(round(kw["T"] / dt) + 1)
    if nt < 3:
        return None
    # the look-back of TunedReward1D indexes row t-100 with Python wrap-around: it raises in the reference when t-100 < -nt
    if S < 100 and nt < 100 - 0 and True:
        pass
    x = np.linspace(0, 1, n)
    style = rng.choice(["smooth", "const", "zeros", "tiny", "big", "compact"])
    if style == "smooth":
        init = rng.uniform(0.5, 5, (B, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (B, 1))))
    elif style == "const":
        init = rng.uniform(-3, 10, (B, 1)) * np.ones((1, n))
    elif style == "zeros":
        init = np.zeros((B, n))
    elif style == "tiny":
        init = rng.uniform(-1, 1, (B, n)) * 1e-38
    elif style == "big":
        init = rng.uniform(1, 2, (B, n)) * 1e30
    else:
        init = np.zeros((B, n))
        init[:, n // 3: n // 3 + 2] = 2.0
    if kind == "burgers":
        init = np.abs(init) * (0.2 if style not in ("big",) else 1.0)
    init = init.astype(np.float32)
    amp = 50 if base == "parabolic" else 5
    shared = bool(rng.random() < 0.3)
    beta = (amp * rng.uniform(0, 1) * np.cos(rng.uniform(1, 8.5, (1 if shared else B, 1)) * np.arccos(x))).astype(np.float32)
    rargs = (nt - 1, float(rng.choice([-1e3, -1e-4])), float(rng.choice([3e2, 1e2])))
    # reward evaluated in the kernel epilogue: TunedReward1D, or NormReward "temporal" with the 1 / 2 / inf norm
    rkind = str(rng.choice(["tuned", "tuned", "tuned", "n1", "n2", "ninf"]))
    rcode = {"tuned": N.REWARD_TUNED1D, "n1": N.REWARD_NORM_L1, "n2": N.REWARD_NORM_L2, "ninf": N.REWARD_NORM_LINF}[rkind]
    # ... or its "differential" horizon on every other NormReward case (no extra draw: the case stream stays what it was)
    # ... and "t-horizon" (mean of the last k row norms, k from the case index) on every fourth
    hz = "temporal" if rkind == "tuned" else ("t-horizon" if idx % 4 == 3 else ("differential" if idx % 2 == 1 else "temporal"))
    hcode = {"temporal": N.HORIZON_TEMPORAL, "differential": N.HORIZON_DIFFERENTIAL, "t-horizon": N.HORIZON_T}[hz]
    klen = (1, 3, 7, 40, 128)[(idx // 4) % 5]
    mk_reward = (lambda: po.TunedReward1DOracle(*rargs)) if rkind == "tuned" else \
        (lambda: po.NormRewardOracle(rargs[0], {"n1": "1", "n2": "2", "ninf": "inf"}[rkind], rargs[1], rargs[2], hz, klen))
    # the reference's mixed-precision modes: float64 plant parameter, float64 / Python-scalar control input
    beta64 = bool(rng.random() < 0.2)
    akind = str(rng.choice(["f32", "f32", "f32", "f64", "weak"]))
    acode = {"f32": N.ACTION_F32, "f64": N.ACTION_F64, "weak": N.ACTION_WEAK}[akind]
    if beta64:
        beta = amp * rng.uniform(0, 1) * np.cos(rng.uniform(1, 8.5, (1 if shared else B, 1)) * np.arccos(x))
    hist = bool(rng.random() < 0.25)
    auto = bool(rng.random() < 0.35)          # fused VecEnv auto-reset from a pool of initial rows
    ocls = {"parabolic": po.ParabolicOracle, "transport": po.TransportOracle, "burgers": po.BurgersOracle}[kind]
    okw = {k: kw[k] for k in ("T", "dt", "X", "dx", "control_sample_rate", "control_type", "sensing_loc", "sensing_type", "normalize",
                              "max_control_value", "limit_pde_state_size", "max_state_value")}
    P = B * int(rng.choice([1, 1, 2, 3]))           # pool rows: the k-th restart of instance b takes row (b + k*B) mod P
    pool = (rng.uniform(0.1, 3, (P, 1)) * np.ones((1, n))).astype(np.float32)
    use_bpool = auto and (not shared) and bool(rng.random() < 0.5)
    bpool = (amp * rng.uniform(0, 1) * np.cos(rng.uniform(1, 8.5, (P, 1)) * np.arccos(x)))
    bpool = bpool if beta64 else bpool.astype(np.float32)
    desc = f"#{idx} P={P} bpool={use_bpool} reward={rkind}/{hz}{klen if hz == 't-horizon' else ''} beta64={beta64} act={akind} auto={auto} {kind} nx={nx} S={S} nt={nt} B={B} {ctrl} {sloc}/{kw['sensing_type']} norm={normalize} limit={limit}/{max_state} ic={style} hist={hist} shared_beta={shared}"
    try:
        orc = ocls(reward=mk_reward(), keep_history=True, **okw)
    except Exception as ex:      # invalid option combination: the product must refuse it too
        try:
            PDEBatch1D(base, reward=RewardSpec(rcode, *rargs, hcode, klen), num_envs=B, device="cuda", flux=flux, **kw)
        except Exception:
            return desc + " (both refuse)"
        raise AssertionError(desc + f": oracle refuses ({ex}) but the engine accepts")
    env = PDEBatch1D(base, reward=RewardSpec(rcode, *rargs, hcode, klen), num_envs=B, device="cuda", record_history=hist, flux=flux, **kw)
    bfull = np.tile(beta, (B, 1)) if shared else beta
    o_ref = orc.reset(init, bfull)
    o_gpu = env.reset(torch.tensor(init), torch.tensor(beta[0] if shared else beta))
    assert np.array_equal(o_gpu.cpu().numpy().reshape(B, -1), np.asarray(o_ref, dtype=np.float32).reshape(B, -1)), desc + " reset obs"
    if auto:
        env.enable_auto_reset(torch.tensor(pool), keep_final_obs=True, beta_pool=torch.tensor(bpool) if use_bpool else None)
        cnt = np.zeros(B, dtype=np.int64)
    for i in range(nsteps + 1):
        a = rng.uniform(-1, 1, B) * float(rng.choice([1.0, 0.0, 10.0]))
        a = a.astype(np.float32) if akind == "f32" else a
        try:
            with np.errstate(all="ignore"):
                o_ref, r_ref, te_ref, tr_ref = orc.step(a, action_kind=akind)
        except IndexError:
            return desc + " (reference look-back raises: skipped)"
        o_gpu, r_gpu, te_gpu, tr_gpu = env.step(torch.tensor(a), action_kind=acode)
        og = o_gpu.cpu().numpy().reshape(B, -1)
        orf = np.asarray(o_ref, dtype=np.float32).reshape(B, -1)
        if auto:
            done = te_ref | tr_ref
            if done.any():      # the kernel restarted these instances: terminal observation kept, new episode's first obs returned
                fo = env.t["final_obs"].cpu().numpy().reshape(B, -1)
                okf = ~(np.isnan(fo) & np.isnan(orf))
                assert np.array_equal(fo[done].view(np.uint32)[okf[done]], orf[done].view(np.uint32)[okf[done]]), desc + f" step {i}: final_obs"
                rows = (np.arange(B) + cnt * B) % P
                cnt[done] += 1
                if use_bpool:
                    orc.beta = orc.beta.copy()
                    orc.beta[done] = bpool[rows[done]]
                fresh = ocls(reward=mk_reward(), keep_history=True, **okw)
                fresh_obs = np.asarray(fresh.reset(pool[rows], orc.beta), dtype=np.float32).reshape(B, -1)
                orc.row[done] = fresh.row[done]
                orc.time_index[done] = 0
                orc.bsum[done] = fresh.bsum[done]
                orc.ring[done, 0] = po._rownorm(fresh.row[done])
                if orc._thor:
                    orc.kring[done, 0] = orc.reward.row_norms(fresh.row[done])
                orc.hist[done] = 0
                orc.hist[done, 0] = fresh.row[done]
                orf = orf.copy()
                orf[done] = fresh_obs[done]
        row_g, row_o = env.u.cpu().numpy(), orc.row
        if not np.array_equal(row_g.view(np.uint32), row_o.view(np.uint32)):
            bad = np.argwhere(row_g.view(np.uint32) != row_o.view(np.uint32))
            # NaN payloads may differ; anything else is a failure
            fin = ~(np.isnan(row_g) & np.isnan(row_o))
            if (row_g.view(np.uint32)[fin] != row_o.view(np.uint32)[fin]).any():
                raise AssertionError(desc + f" step {i}: rows differ at {bad[:4].tolist()} gpu {row_g[tuple(bad[0])]} ref {row_o[tuple(bad[0])]}")
        fin = ~(np.isnan(og) & np.isnan(orf))
        assert np.array_equal(og.view(np.uint32)[fin], orf.view(np.uint32)[fin]), desc + f" step {i}: obs differ"
        assert np.array_equal(env.time_index.cpu().numpy(), orc.time_index), desc + f" step {i}: time index"
        assert np.array_equal(te_gpu.cpu().numpy().astype(bool), te_ref), desc + f" step {i}: terminate"
        assert np.array_equal(tr_gpu.cpu().numpy().astype(bool), tr_ref), desc + f" step {i}: truncate {env.t['norm_now'].cpu().numpy()} {orc.norm_now}"
        rg = r_gpu.cpu().numpy()
        ok = np.isclose(rg, r_ref, rtol=2e-6, atol=4e-6 * max(1.0, float(np.nanmax(np.abs(orc.norm_now)))) if np.isfinite(orc.norm_now).any() else 1.0, equal_nan=True)
        ok |= ~np.isfinite(r_ref) & ~np.isfinite(rg)
        ok |= ~np.isfinite(orc.norm_now) | (np.abs(orc.norm_now) > 1e30)
        assert ok.all(), desc + f" step {i}: reward gpu {rg} ref {r_ref}"
    if hist:
        hg, ho = env.t["history"].cpu().numpy(), orc.hist
        fin = ~(np.isnan(hg) & np.isnan(ho))
        assert np.array_equal(hg.view(np.uint32)[fin], ho.view(np.uint32)[fin]), desc + ": history differs"
    return desc


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, k, skipped = time.time(), 0, 0
    while time.time() - t0 < seconds:
        d = one_case(rng, k)
        k += 1
        if d is None or d.endswith(")"):
            skipped += 1
        if k % 50 == 0:
            print(f"{k} cases ok ({skipped} skipped)  last: {d}", flush=True)
    print(f"FUZZ OK: {k} cases, {skipped} skipped, seed {seed}")
