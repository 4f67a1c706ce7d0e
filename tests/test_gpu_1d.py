"""GPU parity tests for the 1D HIP kernels, called through the C ABI (libpdegym_hip.so).

Bars (stated tolerances):
  * fields / observations: BIT-EXACT float32 vs the reference's golden vectors and vs the oracle
    (kernels are built with -ffp-contract=off and use true IEEE division);
  * norms / rewards: rtol 1e-6 (wave butterfly vs BLAS sdot summation order, float32).
"""
import numpy as np
import pytest

from tests.cases import ACTION_KIND, MIXED_CASES, PARABOLIC_CASES, TRANSPORT_CASES

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _mk(kind, kw, reward_args, B, record_history=False):
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    spec = RewardSpec(N.REWARD_TUNED1D, int(reward_args[0]), float(reward_args[1]), float(reward_args[2]))
    return PDEBatch1D(kind, kw["T"], kw["dt"], kw["X"], kw["dx"], kw["control_sample_rate"],
                      control_type=kw["control_type"], sensing_loc=kw["sensing_loc"], sensing_type=kw["sensing_type"],
                      normalize=kw["normalize"], max_control_value=kw["max_control_value"],
                      limit_pde_state_size=kw["limit_pde_state_size"], max_state_value=kw["max_state_value"],
                      reward=spec, num_envs=B, device="cuda", record_history=record_history)


def _run_golden(kind, kw, g, B=3):
    """Instance 0..B-1 all run the golden case (every instance must reproduce it)."""
    env = _mk(kind, kw, g.reward_args, B)
    init = torch.tensor(np.tile(g.init.astype(np.float32)[None], (B, 1)))
    beta = torch.tensor(np.tile(g.beta.astype(np.float32)[None], (B, 1)))
    obs = env.reset(init, beta)
    np.testing.assert_array_equal(obs.cpu().numpy()[B - 1].reshape(-1), g.obs[0])
    for i, a in enumerate(g.actions):
        obs, r, te, tr = env.step(torch.full((B,), float(a), dtype=torch.float32))
        obs, r, te, tr = obs.cpu().numpy(), r.cpu().numpy(), te.cpu().numpy(), tr.cpu().numpy()
        rows = env.u.cpu().numpy()
        for b in (0, B - 1):
            np.testing.assert_array_equal(rows[b], g.rows[i], err_msg=f"row step {i} inst {b}")
            np.testing.assert_array_equal(obs[b].reshape(-1), g.obs[i + 1], err_msg=f"obs step {i}")
            assert int(env.time_index[b]) == int(g.time_index[i])
            assert bool(te[b]) == bool(g.terminate[i]) and bool(tr[b]) == bool(g.truncate[i]), f"flags step {i}"
            if np.isfinite(g.reward[i]):
                nrm = float(np.linalg.norm(g.rows[i]))
                np.testing.assert_allclose(r[b], g.reward[i], rtol=1e-6, atol=2e-6 * max(1.0, nrm), err_msg=f"reward step {i}")


@pytest.mark.parametrize("case", sorted(TRANSPORT_CASES))
def test_transport_hip_matches_reference_golden(golden_transport, case):
    _run_golden("transport", TRANSPORT_CASES[case], golden_transport[case])


@pytest.mark.parametrize("case", sorted(PARABOLIC_CASES))
def test_parabolic_hip_matches_reference_golden(golden_parabolic, case):
    _run_golden("parabolic", PARABOLIC_CASES[case], golden_parabolic[case])


@pytest.mark.parametrize("case", sorted(MIXED_CASES))
def test_mixed_precision_hip_matches_reference_golden(golden_mixed, case):
    """float64 beta and/or float64 / Python-scalar control (the reference's mixed-precision arithmetic, incl. the docs
    quickstart 'Q_quick'): rows and observations BIT-EXACT on the HIP path, engine level (dtype of the beta tensor and the
    action kind select the kernel mode)."""
    from pdecontrolgym_amd import _native as N
    kind, kw, action_as, _, _ = MIXED_CASES[case]
    g = golden_mixed[case]
    B = 3
    env = _mk(kind, kw, g.reward_args, B)
    init = torch.tensor(np.tile(g.init.astype(np.float32)[None], (B, 1)))
    beta = torch.tensor(np.tile(g.beta[None], (B, 1)))          # dtype preserved: float64 / int64 / float32
    obs = env.reset(init, beta)
    assert env.params.beta_f64 == (0 if g.beta.dtype == np.float32 else 1)
    np.testing.assert_array_equal(obs.cpu().numpy()[B - 1].reshape(-1), g.obs[0])
    ak = {"f32": N.ACTION_F32, "f64": N.ACTION_F64, "weak": N.ACTION_WEAK}[ACTION_KIND[action_as]]
    for i, a in enumerate(g.actions):
        obs, r, te, tr = env.step(torch.full((B,), float(a), dtype=torch.float32 if ak == N.ACTION_F32 else torch.float64), action_kind=ak)
        obs, r, te, tr = obs.cpu().numpy(), r.cpu().numpy(), te.cpu().numpy(), tr.cpu().numpy()
        rows = env.u.cpu().numpy()
        for b in (0, B - 1):
            np.testing.assert_array_equal(rows[b], g.rows[i], err_msg=f"row step {i} inst {b}")
            np.testing.assert_array_equal(obs[b].reshape(-1), g.obs[i + 1], err_msg=f"obs step {i}")
            assert int(env.time_index[b]) == int(g.time_index[i])
            assert bool(te[b]) == bool(g.terminate[i]) and bool(tr[b]) == bool(g.truncate[i]), f"flags step {i}"
            if np.isfinite(g.reward[i]):
                nrm = float(np.linalg.norm(g.rows[i]))
                np.testing.assert_allclose(r[b], g.reward[i], rtol=1e-6, atol=2e-6 * max(1.0, nrm), err_msg=f"reward step {i}")


def _oracle_kwargs(kw):
    return {k: kw[k] for k in ("T", "dt", "X", "dx", "control_sample_rate", "control_type", "sensing_loc",
                               "sensing_type", "normalize", "max_control_value", "limit_pde_state_size",
                               "max_state_value")}


@pytest.mark.parametrize("kind,nx,S,B,ctrl", [
    ("parabolic", 256, 100, 64, "Dirchilet"), ("parabolic", 256, 7, 33, "Neumann"), ("parabolic", 200, 100, 16, "Dirchilet"),
    ("parabolic", 63, 10, 9, "Dirchilet"), ("parabolic", 640, 20, 5, "Dirchilet"), ("parabolic", 1000, 5, 3, "Neumann"),
    ("transport", 512, 100, 64, "Dirchilet"), ("transport", 100, 50, 17, "Neumann"), ("transport", 64, 10, 8, "Dirchilet"),
    ("transport", 65, 10, 8, "Dirchilet"), ("transport", 300, 25, 6, "Dirchilet"), ("transport", 1024, 12, 4, "Dirchilet"),
    ("transport", 31, 12, 5, "Dirchilet"), ("parabolic", 2, 3, 4, "Dirchilet"),
    ("transport", 1500, 9, 3, "Dirchilet"), ("parabolic", 2047, 4, 3, "Dirchilet"), ("transport", 2048, 5, 2, "Neumann"),
    ("parabolic", 1300, 6, 2, "Neumann"),
    # rows beyond 2048 nodes: the LDS-resident wide kernel
    ("transport", 2049, 7, 3, "Dirchilet"), ("parabolic", 3000, 5, 2, "Neumann"), ("transport", 4096, 4, 2, "Neumann"),
    ("parabolic", 8191, 3, 2, "Dirchilet"),
])
def test_hip_matches_oracle_random_batches(kind, nx, S, B, ctrl):
    """Seeded random per-instance IC / beta / actions: every row bit-exact vs the oracle."""
    from oracle import pde_oracle as po
    rng = np.random.default_rng(nx * 1000 + S)
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    nsteps = 6
    kw = dict(T=(nsteps - 1) * S * dt + 3 * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=ctrl,
              sensing_loc="full", sensing_type=None, normalize=False, max_control_value=20,
              limit_pde_state_size=True, max_state_value=1e10)
    n = nx + (1 if kind == "parabolic" else 0)
    x = np.linspace(0, 1, n)
    amp = 50 if kind == "parabolic" else 5
    beta = (amp * np.cos(rng.uniform(7, 8.5, (B, 1)) * np.arccos(x))).astype(np.float32)
    init = (rng.uniform(1, 10, (B, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (B, 1))))).astype(np.float32)
    rargs = (int(round(kw["T"] / dt)), -1e3, 3e2)
    cls = po.ParabolicOracle if kind == "parabolic" else po.TransportOracle
    orc = cls(reward=po.TunedReward1DOracle(*rargs), keep_history=False, **_oracle_kwargs(kw))
    env = _mk(kind, kw, rargs, B)
    o_ref = orc.reset(init, beta)
    o_gpu = env.reset(torch.tensor(init), torch.tensor(beta))
    np.testing.assert_array_equal(o_gpu.cpu().numpy(), o_ref)
    for i in range(nsteps + 1):          # the last call is post-terminal (0 sub-steps)
        a = rng.uniform(-1, 1, B).astype(np.float32)
        with np.errstate(all="ignore"):
            o_ref, r_ref, te_ref, tr_ref = orc.step(a)
        o_gpu, r_gpu, te_gpu, tr_gpu = env.step(torch.tensor(a))
        np.testing.assert_array_equal(env.u.cpu().numpy(), orc.row, err_msg=f"step {i}")
        np.testing.assert_array_equal(o_gpu.cpu().numpy(), o_ref)
        np.testing.assert_array_equal(env.time_index.cpu().numpy(), orc.time_index)
        np.testing.assert_array_equal(te_gpu.cpu().numpy().astype(bool), te_ref)
        np.testing.assert_array_equal(tr_gpu.cpu().numpy().astype(bool), tr_ref)
        np.testing.assert_allclose(env.t["norm_now"].cpu().numpy(), orc.norm_now, rtol=1e-6)
        np.testing.assert_allclose(r_gpu.cpu().numpy(), r_ref, rtol=1e-6, atol=2e-6 * float(np.max(orc.norm_now)))


@pytest.mark.parametrize("kind,nx,S", [("parabolic", 256, 100), ("parabolic", 100, 40), ("transport", 128, 50), ("transport", 100, 30)])
def test_denormal_and_compact_support_states_match_numpy_bitwise(kind, nx, S):
    """float32 denormals are kept, not flushed: (a) a compactly supported initial condition whose diffusion front decays
    through the denormal range to zero, (b) rows that live entirely in the denormal range, (c) zero actions / zero beta
    entries.  NumPy on the CPU honours denormals; every row must still be bit-identical."""
    from oracle import pde_oracle as po
    rng = np.random.default_rng(nx + S)
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    nsteps, B = 4, 6
    kw = dict(T=nsteps * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet",
              sensing_loc="full", sensing_type=None, normalize=False, max_control_value=20,
              limit_pde_state_size=True, max_state_value=1e10)
    n = nx + (1 if kind == "parabolic" else 0)
    init = np.zeros((B, n), dtype=np.float32)
    init[0, n // 2 - 2: n // 2 + 2] = 3.0                                   # compact support: the front underflows
    init[1, : n // 3] = rng.uniform(1, 2, n // 3)
    init[2] = (rng.uniform(-1, 1, n) * 1e-39).astype(np.float32)            # everything denormal
    init[3] = (rng.uniform(0.5, 1, n) * 2.0 ** -120).astype(np.float32)     # just above the denormal range, decaying into it
    init[4, ::7] = np.float32(1e-44)
    init[5] = rng.uniform(1, 2, n)
    assert (np.abs(init[2][init[2] != 0]) < 1.2e-38).all()
    beta = np.zeros((B, n), dtype=np.float32)
    beta[5] = rng.uniform(-1, 1, n)
    beta[3] = -30.0 if kind == "parabolic" else 0.0
    rargs = (int(round(kw["T"] / dt)), -1e3, 3e2)
    cls = po.ParabolicOracle if kind == "parabolic" else po.TransportOracle
    orc = cls(reward=po.TunedReward1DOracle(*rargs), keep_history=False, **_oracle_kwargs(kw))
    env = _mk(kind, kw, rargs, B)
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    saw_denormal = False
    for i in range(nsteps):
        a = np.array([0.0, 0.0, 1e-40, 0.0, 0.0, 0.3], dtype=np.float32)
        with np.errstate(all="ignore"):
            orc.step(a)
        env.step(torch.tensor(a))
        row = orc.row
        saw_denormal |= bool(((np.abs(row) < 1.17e-38) & (row != 0)).any())
        np.testing.assert_array_equal(env.u.cpu().numpy().view(np.uint32), row.view(np.uint32), err_msg=f"step {i}")
    assert saw_denormal


@pytest.mark.parametrize("nx,S,B,ctrl", [(512, 100, 16, "Dirchilet"), (100, 25, 7, "Neumann"), (65, 30, 5, "Dirchilet"), (1500, 30, 2, "Dirchilet")])
def test_burgers_extension_matches_own_restatement(nx, S, B, ctrl):
    """EXTENSION, parity unpinned (the reference has no Burgers environment): the u u_x flux mode of the transport kernel
    against this repository's NumPy restatement, bit for bit; power-of-two and other dx, both control types, history."""
    from oracle import pde_oracle as po
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    rng = np.random.default_rng(nx + S)
    dx = 1.0 / nx
    dt = 0.5 * dx
    nsteps = 5
    kw = dict(T=nsteps * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=ctrl, sensing_loc="full",
              sensing_type=None, normalize=False, max_control_value=20, limit_pde_state_size=True, max_state_value=1e10)
    x = np.linspace(0, 1, nx)
    init = (rng.uniform(0.2, 1.0, (B, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (B, 1))))).astype(np.float32)
    beta = (0.2 * (1 + np.cos(rng.uniform(2, 4, (B, 1)) * x))).astype(np.float32)          # non-negative: u stays positive
    rargs = (int(round(kw["T"] / dt)), -1e3, 3e2)
    orc = po.BurgersOracle(reward=po.TunedReward1DOracle(*rargs), keep_history=True, **_oracle_kwargs(kw))
    env = PDEBatch1D("transport", reward=RewardSpec(N.REWARD_TUNED1D, *rargs), num_envs=B, device="cuda", record_history=(nx == 65),
                     flux="burgers", **{k: kw[k] for k in kw})
    lin = PDEBatch1D("transport", reward=RewardSpec(N.REWARD_TUNED1D, *rargs), num_envs=B, device="cuda", **{k: kw[k] for k in kw})
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    lin.reset(torch.tensor(init), torch.tensor(beta))
    for i in range(nsteps):
        a = rng.uniform(0, 1, B).astype(np.float32)
        with np.errstate(all="ignore"):
            o_ref, r_ref, te_ref, tr_ref = orc.step(a)
        o_gpu, r_gpu, te_gpu, tr_gpu = env.step(torch.tensor(a))
        lin.step(torch.tensor(a))
        np.testing.assert_array_equal(env.u.cpu().numpy().view(np.uint32), orc.row.view(np.uint32), err_msg=f"step {i}")
        assert np.isfinite(orc.row).all()
        np.testing.assert_allclose(r_gpu.cpu().numpy(), r_ref, rtol=1e-6, atol=2e-6 * float(np.max(orc.norm_now)))
        np.testing.assert_array_equal(te_gpu.cpu().numpy().astype(bool), te_ref)
    assert not torch.equal(env.u, lin.u)          # the flux really differs from the reference's linear transport
    if nx == 65:
        np.testing.assert_array_equal(env.t["history"].cpu().numpy(), orc.hist)


@pytest.mark.parametrize("horizon", ["temporal", "differential", "t-horizon:5", "t-horizon:25", "t-horizon:1"])
@pytest.mark.parametrize("norm", ["1", "2", "inf"])
@pytest.mark.parametrize("kind,nx,beta64", [("parabolic", 256, False), ("transport", 100, False), ("transport", 2100, False),
                                            ("parabolic", 2500, False), ("parabolic", 200, True), ("transport", 64, True)])
def test_norm_reward_epilogues_match_oracle(kind, nx, beta64, norm, horizon):
    """PDEGYM_REWARD_NORM_L1 / L2 / LINF (NormReward, parity unpinned: the reference class raises) through the
    register-resident kernel (n <= 2048), the wide LDS kernel (n > 2048) and the mixed-precision kernel (float64 beta) against
    NormRewardOracle: -||u_t|| ("temporal") or +||u_t - u_{t-1}|| over fine-time rows ("differential", evaluated by the
    select-form kernel, which keeps the row before its last sub-step) or -(mean of the norms of the last k fine-time rows)
    ("t-horizon:k": k = 5 inside one env-step of 10 sub-steps, 25 across three, 1 = "temporal") per step, the truncation penalty and
    the terminal reward.  rtol 1e-6 (reduction order); the differential norm is a sum of rounded differences, so atol 1e-6 of the
    row scale."""
    from oracle import pde_oracle as po
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    rng = np.random.default_rng(nx + int(beta64))
    S, nsteps, B = 10, 5, 6
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    kw = dict(T=nsteps * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=False, max_control_value=20, limit_pde_state_size=True, max_state_value=1e10)
    n = nx + (1 if kind == "parabolic" else 0)
    x = np.linspace(0, 1, n)
    beta = (5 * np.cos(rng.uniform(7, 8, (B, 1)) * np.arccos(x)))
    beta = beta if beta64 else beta.astype(np.float32)
    init = (rng.uniform(2, 10, (B, 1)) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3, (B, 1))))).astype(np.float32)
    init[3] = -init[3]
    # instance 0 stays above max_state_value, the others well below -> the truncation branch is taken by instance 0 only
    rargs = (int(round(kw["T"] / dt)), -2.5, 77.0)
    code = {"1": N.REWARD_NORM_L1, "2": N.REWARD_NORM_L2, "inf": N.REWARD_NORM_LINF}[norm]
    l2_0 = float(np.linalg.norm(init[0]))
    kw["max_state_value"] = 0.5 * l2_0
    init[1:] *= np.float32(0.2 * l2_0 / np.max(np.linalg.norm(init[1:], axis=1)))
    horizon, _, klen = horizon.partition(":")
    klen = int(klen or 5)
    orc = (po.ParabolicOracle if kind == "parabolic" else po.TransportOracle)(
        reward=po.NormRewardOracle(rargs[0], norm, rargs[1], rargs[2], horizon, klen), keep_history=False, **_oracle_kwargs(kw))
    hz = {"temporal": N.HORIZON_TEMPORAL, "differential": N.HORIZON_DIFFERENTIAL, "t-horizon": N.HORIZON_T}[horizon]
    env = PDEBatch1D(kind, reward=RewardSpec(code, *rargs, hz, klen), num_envs=B, device="cuda", **kw)
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    assert env.can_rollout() == (horizon == "temporal" and not beta64 and n <= N.MAX_N1D_REG)
    saw_trunc = saw_term = False
    for i in range(nsteps):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        with np.errstate(all="ignore"):
            o_ref, r_ref, te_ref, tr_ref = orc.step(a)
        o, r, te, tr = env.step(torch.tensor(a))
        np.testing.assert_array_equal(env.u.cpu().numpy(), orc.row, err_msg=f"step {i}")
        np.testing.assert_array_equal(te.cpu().numpy().astype(bool), te_ref)
        np.testing.assert_array_equal(tr.cpu().numpy().astype(bool), tr_ref)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-6, err_msg=f"step {i}")
        saw_trunc |= bool((tr_ref & ~te_ref).any())
        saw_term |= bool(te_ref.any())
        live = r_ref[~te_ref & ~tr_ref]
        assert (live > 0).all() if horizon == "differential" else (live < 0).all()
    assert saw_trunc and saw_term


def test_history_recording_matches_oracle():
    from oracle import pde_oracle as po
    kw = dict(PARABOLIC_CASES["P2_s1"])
    B, n = 2, 257
    rng = np.random.default_rng(5)
    init = rng.uniform(1, 3, (B, n)).astype(np.float32)
    beta = rng.uniform(-10, 10, (B, n)).astype(np.float32)
    orc = po.ParabolicOracle(reward=po.TunedReward1DOracle(1000, -1e3, 3e2), keep_history=True, **_oracle_kwargs(kw))
    env = _mk("parabolic", kw, (1000, -1e3, 3e2), B, record_history=True)
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    for i in range(20):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        orc.step(a)
        env.step(torch.tensor(a))
    np.testing.assert_array_equal(env.t["history"].cpu().numpy(), orc.hist)


def test_masked_reset_and_shared_beta():
    from oracle import pde_oracle as po
    kw = dict(TRANSPORT_CASES["H1"])
    B, n = 6, 100
    rng = np.random.default_rng(11)
    beta = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, n)))).astype(np.float32)
    init = rng.uniform(1, 10, (B, n)).astype(np.float32)
    env = _mk("transport", kw, (10000, -1e3, 3e2), B)
    env.reset(torch.tensor(init), torch.tensor(beta))          # 1-D beta = shared row (stride 0)
    a = rng.uniform(-1, 1, B).astype(np.float32)
    env.step(torch.tensor(a))
    env.step(torch.tensor(a))
    mask = np.array([1, 0, 0, 1, 0, 1], dtype=np.uint8)
    init2 = rng.uniform(1, 10, (B, n)).astype(np.float32)
    obs = env.reset(torch.tensor(init2), mask=torch.tensor(mask))
    ti = env.time_index.cpu().numpy()
    np.testing.assert_array_equal(ti, np.where(mask, 0, 2000))
    np.testing.assert_array_equal(env.u.cpu().numpy()[mask.astype(bool)], init2[mask.astype(bool)])
    np.testing.assert_array_equal(obs.cpu().numpy()[mask.astype(bool)], init2[mask.astype(bool)])
    # continue: reset instances behave like fresh oracles, the others like 3-step-old ones
    a3 = rng.uniform(-1, 1, B).astype(np.float32)
    env.step(torch.tensor(a3))
    kwargs = _oracle_kwargs(kw)
    for b in range(B):
        orc = po.TransportOracle(reward=po.TunedReward1DOracle(10000, -1e3, 3e2), keep_history=False, **kwargs)
        if mask[b]:
            orc.reset(init2[b:b + 1], beta[None])
        else:
            orc.reset(init[b:b + 1], beta[None])
            orc.step(a[b:b + 1])
            orc.step(a[b:b + 1])
        _, r, _, _ = orc.step(a3[b:b + 1])
        np.testing.assert_array_equal(env.u.cpu().numpy()[b], orc.row[0])
        np.testing.assert_allclose(env.t["reward"].cpu().numpy()[b], r[0], rtol=1e-6, atol=1e-4)


def test_baseline_size_properties_c2():
    """BASELINE config 2 (nx=256, B=4096, S=100): properties that need no oracle at full size:
    (1) batch invariance -- duplicated instances give identical rows; (2) u(0,t) == 0 and the boundary
    node equals the action; (3) linearity of the (linear) plant in the state for zero action; plus 64 sampled instances
    bit-identical to the oracle (rows) / rtol 1e-6 (rewards)."""
    from pdecontrolgym_amd import _native as N
    B, n = 4096, 257
    dx = 1.0 / 256
    dt = 0.25 * dx * dx
    kw = dict(T=1000 * 100 * dt, dt=dt, X=1, dx=dx, control_sample_rate=100 * dt, control_type="Dirchilet",
              sensing_loc="full", sensing_type=None, normalize=False, max_control_value=20,
              limit_pde_state_size=True, max_state_value=1e10)
    from oracle import pde_oracle as po
    env = _mk("parabolic", kw, (100000, -1e3, 3e2), B)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.linspace(0, 1, n)
    gam = torch.rand(B // 2, 1, generator=g) + 7.5
    beta = (50 * torch.cos(gam * torch.acos(x))).float()
    beta = torch.cat([beta, beta])
    init = (torch.rand(B // 2, 1, generator=g) * 9 + 1) * torch.ones(1, n)
    init = torch.cat([init, init]).float()
    env.reset(init, beta)
    # 64 instances spread over the batch also run on the oracle: rows bit-identical, rewards rtol 1e-6, at the FULL batch size
    sel = torch.arange(0, B, B // 64)
    orc = po.ParabolicOracle(reward=po.TunedReward1DOracle(100000, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
    orc.reset(init[sel].numpy(), beta[sel].numpy())
    for _ in range(3):
        a = torch.rand(B // 2, generator=g) * 2 - 1
        a = torch.cat([a, a])
        _, r, _, _ = env.step(a)
        _, r_ref, _, _ = orc.step(a[sel].numpy())
        np.testing.assert_array_equal(env.u.cpu()[sel].numpy(), orc.row)
        np.testing.assert_allclose(r.cpu()[sel].numpy(), r_ref, rtol=1e-6, atol=2e-6 * float(np.max(orc.norm_now)))
    u = env.u.cpu()
    assert torch.equal(u[: B // 2], u[B // 2:])
    assert torch.all(u[:, 0] == 0) and torch.equal(u[:, -1], a)
    assert torch.isfinite(u).all()
    # scaling by a power of two is exact in binary floating point -> exact linearity check
    env2 = _mk("parabolic", kw, (100000, -1e3, 3e2), B)
    env2.reset(init * 4, beta)
    env.reset(init, beta)
    z = torch.zeros(B)
    env.step(z)
    env2.step(z)
    assert torch.equal(env.u.cpu() * 4, env2.u.cpu())
    # checksum of row norms == norm_now
    nn = env.t["norm_now"].cpu()
    torch.testing.assert_close(nn, torch.linalg.vector_norm(env.u.cpu().double(), dim=1).float(), rtol=1e-6, atol=0)


def test_baseline_size_properties_c3():
    """BASELINE config 3 shape (transport nx=512, B=16384, S=100; dx = 2^-9 takes the one-multiply quotient): duplicated
    instances agree, the controlled node equals the action, exact linearity under a power-of-two scaling of state AND
    action, norm checksum, and 64 sampled instances bit-identical to the oracle."""
    from oracle import pde_oracle as po
    B, n = 16384, 512
    dx = 1.0 / 512
    dt = 0.5 * dx
    kw = dict(T=1000 * 100 * dt, dt=dt, X=1, dx=dx, control_sample_rate=100 * dt, control_type="Dirchilet",
              sensing_loc="full", sensing_type=None, normalize=False, max_control_value=20,
              limit_pde_state_size=True, max_state_value=1e10)
    rargs = (100000, -1e3, 3e2)
    env = _mk("transport", kw, rargs, B)
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.linspace(0, 1, n)
    gam = torch.rand(B // 2, 1, generator=g) * 0.7 + 7.0
    beta = (5 * torch.cos(gam * torch.acos(x))).float()
    beta = torch.cat([beta, beta])
    init = ((torch.rand(B // 2, 1, generator=g) * 9 + 1) * torch.ones(1, n)).float()
    init = torch.cat([init, init])
    env.reset(init, beta)
    acts = []
    for _ in range(2):
        a = torch.rand(B // 2, generator=g) * 2 - 1
        a = torch.cat([a, a])
        acts.append(a)
        env.step(a)
    u = env.u.cpu()
    assert torch.equal(u[: B // 2], u[B // 2:]) and torch.equal(u[:, -1], a) and torch.isfinite(u).all()
    nn = env.t["norm_now"].cpu()
    torch.testing.assert_close(nn, torch.linalg.vector_norm(u.double(), dim=1).float(), rtol=1e-6, atol=0)
    env2 = _mk("transport", kw, rargs, B)
    env2.reset(init * 8, beta)
    for a in acts:
        env2.step(a * 8)
    assert torch.equal(u * 8, env2.u.cpu())
    sel = torch.arange(0, B, B // 64)
    orc = po.TransportOracle(reward=po.TunedReward1DOracle(*rargs), keep_history=False, **_oracle_kwargs(kw))
    orc.reset(init[sel].numpy(), beta[sel].numpy())
    for a in acts:
        orc.step(a[sel].numpy())
    np.testing.assert_array_equal(u[sel].numpy(), orc.row)


def test_abi_rejects_bad_arguments():
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    with pytest.raises(N.NativeError):
        env = PDEBatch1D("transport", 1, 1e-3, 1, 1.0 / 9000, 0.01, num_envs=2, device="cuda")   # n = 9000 > 8192
        env.step(torch.zeros(2))


def test_abi_rejects_bad_round2_arguments():
    """The ABI fields added in round 2 are validated on the C side: unknown action kind, dtype / flag mismatches caught by the
    binding, a pressure ping-pong buffer that aliases p, auto-reset pools given only in part."""
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    from pdecontrolgym_amd.batch2d import NSBatch2D
    from tests.cases import NS_BC
    env = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda")
    env.reset(torch.ones(2, 100), torch.ones(2, 100))
    with pytest.raises(N.NativeError, match="action_kind"):
        env.step(torch.zeros(2, dtype=torch.float64), action_kind=7)
    env.params.beta_f64 = 1                       # flag says float64 but the tensor is float32
    with pytest.raises(N.NativeError, match="beta"):
        env.step(torch.zeros(2))
    env.params.beta_f64 = 0
    n = 16
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    ns = NSBatch2D(4 * dt, dt, 1, dx, 1, dx, NS_BC, np.zeros((4, n, n, 2)), 2.0 * np.ones(4), num_envs=2, device="cuda", dtype=torch.float64)
    ns.reset(np.zeros((2, n, n)), np.zeros((2, n, n)), np.zeros((2, n, n)))
    ns.t["p_out"] = ns.t["p"]
    with pytest.raises(N.NativeError, match="alias"):
        ns.step(torch.ones(2, 1, dtype=torch.float64))
    ns.t["p_out"] = None
    ns.t["reset_u0"] = torch.zeros(2, n, n, dtype=torch.float64, device="cuda")
    with pytest.raises(N.NativeError, match="reset_v0"):
        ns.step(torch.ones(2, 1, dtype=torch.float64))


def test_fused_auto_reset():
    """Instances that finish restart from the pool inside the launch; others are untouched."""
    from oracle import pde_oracle as po
    kw = dict(TRANSPORT_CASES["R_s30"])                 # nt=401, S=30 -> terminates at step 14
    B, n = 5, 100
    rng = np.random.default_rng(3)
    beta = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, n)))).astype(np.float32)
    init = rng.uniform(1, 3, (B, n)).astype(np.float32)
    pool = rng.uniform(1, 3, (B, n)).astype(np.float32)
    env = _mk("transport", kw, (400, -1e3, 3e2), B)
    env.reset(torch.tensor(init), torch.tensor(beta))
    env.enable_auto_reset(torch.tensor(pool))
    orc = po.TransportOracle(reward=po.TunedReward1DOracle(400, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
    orc.reset(init, np.tile(beta, (B, 1)))
    for i in range(14):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        o_ref, r_ref, te_ref, tr_ref = orc.step(a)
        o, r, te, tr = env.step(torch.tensor(a))
    assert te.cpu().numpy().all() and te_ref.all()
    np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-6, atol=1e-4)
    np.testing.assert_array_equal(env.t["final_obs"].cpu().numpy(), o_ref)      # terminal observation kept
    np.testing.assert_array_equal(o.cpu().numpy(), pool)                        # first obs of the new episode
    np.testing.assert_array_equal(env.u.cpu().numpy(), pool)
    assert (env.time_index.cpu().numpy() == 0).all()
    # the new episode evolves like a fresh oracle
    orc.reset(pool, np.tile(beta, (B, 1)))
    for i in range(3):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        o_ref, r_ref, _, _ = orc.step(a)
        o, r, te, tr = env.step(torch.tensor(a))
        np.testing.assert_array_equal(o.cpu().numpy(), o_ref)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-6, atol=1e-4)


def test_fused_auto_reset_redraws_beta_and_rotates_pool_rows():
    """The reference calls both reset callbacks at every reset (hyperbolic.py:207-209): with pools of P = 3B rows the k-th
    restart of instance b takes initial condition AND beta from row (b + k*B) mod P, inside the launch; every episode
    equals a fresh oracle started from that row."""
    from oracle import pde_oracle as po
    kw = dict(TRANSPORT_CASES["R_tiny"])                # nt=91, S=15 -> an episode is 6 steps
    B, n, P = 4, 100, 12
    rng = np.random.default_rng(8)
    x = np.linspace(0, 1, n)
    pool = rng.uniform(1, 3, (P, n)).astype(np.float32)
    bpool = (5 * np.cos(rng.uniform(7, 8, (P, 1)) * np.arccos(x))).astype(np.float32)
    init = rng.uniform(1, 3, (B, n)).astype(np.float32)
    beta0 = (5 * np.cos(7.35 * np.arccos(x))).astype(np.float32)
    env = _mk("transport", kw, (90, -1e3, 3e2), B)
    env.reset(torch.tensor(init), torch.tensor(beta0))            # shared beta row: enable_auto_reset un-shares it
    env.enable_auto_reset(torch.tensor(pool), beta_pool=torch.tensor(bpool))
    orcs = []
    for b in range(B):
        o = po.TransportOracle(reward=po.TunedReward1DOracle(90, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
        o.reset(init[b:b + 1], beta0[None])
        orcs.append(o)
    restarts = np.zeros(B, dtype=int)
    for i in range(20):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        obs, r, te, tr = env.step(torch.tensor(a))
        obs, r = obs.cpu().numpy(), r.cpu().numpy()
        for b in range(B):
            o_ref, r_ref, te_ref, tr_ref = orcs[b].step(a[b:b + 1])
            np.testing.assert_allclose(r[b], r_ref[0], rtol=1e-6, atol=1e-4)
            if te_ref[0] or tr_ref[0]:
                row = (b + restarts[b] * B) % P
                restarts[b] += 1
                np.testing.assert_array_equal(env.t["final_obs"].cpu().numpy()[b], o_ref[0])
                orcs[b] = po.TransportOracle(reward=po.TunedReward1DOracle(90, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
                o_ref = orcs[b].reset(pool[row:row + 1], bpool[row:row + 1])
                np.testing.assert_array_equal(env.t["beta"].cpu().numpy()[b], bpool[row])
            np.testing.assert_array_equal(obs[b], np.asarray(o_ref)[0], err_msg=f"step {i} inst {b}")
    assert restarts.min() >= 3 and (env.t["reset_count"].cpu().numpy() == restarts).all()


def test_parabolic_transient_overflow_takes_the_exact_loop():
    """|u| >= 2^127: the reference's 2*u overflows to inf (and the row turns NaN for good) while a fused um - 2u would stay
    finite and could decay back below FLT_MAX unnoticed.  The fast loop must not be used there: bit patterns equal the oracle."""
    from oracle import pde_oracle as po
    nx, S, B = 64, 8, 4
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    kw = dict(T=4 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=False, max_control_value=20, limit_pde_state_size=False, max_state_value=1e10)
    n = nx + 1
    init = np.zeros((B, n), dtype=np.float32)
    init[0, 30] = 2.0e38                      # a lone spike above 2^127: the reference overflows in 2*u, the spike itself would halve
    init[1, 10:50] = 1.0e38                   # plateau below 2^127: stays finite in both
    init[2, 20] = -3.0e38
    init[3] = 1.0
    beta = np.zeros((B, n), dtype=np.float32)
    orc = po.ParabolicOracle(reward=po.TunedReward1DOracle(4 * S, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
    env = _mk("parabolic", kw, (4 * S, -1e3, 3e2), B)
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    for i in range(3):
        a = np.zeros(B, dtype=np.float32)
        with np.errstate(all="ignore"):
            orc.step(a)
        env.step(torch.tensor(a))
        g, o = env.u.cpu().numpy(), orc.row
        both_nan = np.isnan(g) & np.isnan(o)
        np.testing.assert_array_equal(g.view(np.uint32)[~both_nan], o.view(np.uint32)[~both_nan], err_msg=f"step {i}")
        np.testing.assert_array_equal(np.isnan(g), np.isnan(o))
    assert np.isnan(orc.row[0]).any() and np.isfinite(orc.row[1]).all() and np.isfinite(orc.row[3]).all()


def test_engine_on_a_non_current_device():
    """An engine built on cuda:1 while cuda:0 is the current device launches on ITS device (the C ABI launches on
    the thread's current device).  Needs two GPUs; skipped on a single-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from oracle import pde_oracle as po
    kw = dict(PARABOLIC_CASES["P2_s100"])
    B, n = 8, 257
    rng = np.random.default_rng(2)
    init = rng.uniform(1, 3, (B, n)).astype(np.float32)
    beta = rng.uniform(-10, 10, (B, n)).astype(np.float32)
    torch.cuda.set_device(0)
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    env = PDEBatch1D("parabolic", reward=RewardSpec(N.REWARD_TUNED1D, 1000, -1e3, 3e2), num_envs=B, device="cuda:1", **kw)
    orc = po.ParabolicOracle(reward=po.TunedReward1DOracle(1000, -1e3, 3e2), keep_history=False, **_oracle_kwargs(kw))
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    for _ in range(3):
        a = rng.uniform(-1, 1, B).astype(np.float32)
        orc.step(a)
        env.step(torch.tensor(a))
    assert torch.cuda.current_device() == 0 and env.u.device.index == 1
    np.testing.assert_array_equal(env.u.cpu().numpy(), orc.row)


@pytest.mark.parametrize("dx", [1e-2, 1.0 / 512, 1.0 / 3, 0.1, 5e-3, 1.0 / 100, 0.999999, 1.9999999, 3.0e-5, 7.0])
def test_transport_quotient_equals_ieee_division(dx):
    """(float)((double)d * (1/(double)dx)) == d / dx for every float32 d: 64 M random bit patterns (all exponents,
    subnormals, infinities, NaNs) + structured values, per dx."""
    import ctypes as C
    from pdecontrolgym_amd import _native as N
    lib = N.load()
    dx32 = float(C.c_float(dx).value)
    n = 1 << 26
    g = torch.Generator(device="cuda").manual_seed(int(dx * 1e6) % 100003)
    bits = torch.randint(-2**31, 2**31 - 1, (n,), generator=g, device="cuda", dtype=torch.int32)
    a = bits.view(torch.float32)
    # structured: exact multiples of dx, powers of two, boundaries
    k = torch.arange(-4096, 4096, device="cuda", dtype=torch.float32)
    extra = torch.cat([k * dx32, k * dx32 * (1 + 2.0 ** -23), 2.0 ** torch.arange(-149, 128, device="cuda", dtype=torch.float32),
                       torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 3.4028234663852886e38, 1.1754944e-38, 1e-45], device="cuda")])
    for arr in (a, extra.contiguous()):
        mism = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = lib.pdegym_selftest_quotient(arr.data_ptr(), dx32, 1.0 / dx32, mism.data_ptr(), arr.numel(), N.current_stream_ptr())
        assert rc == 0
        assert int(mism.item()) == 0


@pytest.mark.parametrize("kind,nx,S,beta64", [("parabolic", 256, 100, False), ("transport", 100, 30, False), ("transport", 3000, 4, False),
                                              ("parabolic", 64, 10, True), ("transport", 700, 3, True)])
def test_state_in_observation_mode_equals_separate_state(kind, nx, S, beta64):
    """Full-state sensing: the rows live in the double-buffered observation tensors (bufs.state_in: one row store per env-step
    instead of two).  Bit for bit the separate-state engine across episode ends with the fused auto-reset (initial-condition and
    beta pools), a masked reset in between, the register / LDS-resident / mixed-precision kernels; the observation returned by
    step k is still intact after step k+1."""
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    B = 7
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    nsteps = 9
    kw = dict(T=5 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=True, max_control_value=5.0, limit_pde_state_size=True, max_state_value=1e6)
    rng = np.random.default_rng(nx + S)
    envs = []
    for mode in (True, False):
        e = PDEBatch1D(kind, reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B, device="cuda",
                       state_in_obs=mode, **kw)
        assert e.state_in_obs == mode
        envs.append(e)
    n = envs[0].n
    bdt = np.float64 if beta64 else np.float32
    init = rng.uniform(1, 3, (B, n)).astype(np.float32)
    beta = rng.uniform(-2, 2, (B, n)).astype(bdt)
    pool_i = rng.uniform(1, 3, (3 * B, n)).astype(np.float32)
    pool_b = rng.uniform(-2, 2, (3 * B, n)).astype(bdt)
    acts = rng.uniform(-1, 1, (nsteps, B)).astype(np.float32)
    outs = []
    for e in envs:
        o0 = e.reset(torch.tensor(init), torch.tensor(beta))
        e.enable_auto_reset(torch.tensor(pool_i), keep_final_obs=True, beta_pool=torch.tensor(pool_b))
        res = [o0.cpu().numpy().copy()]
        prev, prev_copy = None, None
        for k in range(nsteps):
            if k == 4:
                m = torch.tensor([1, 0, 0, 1, 0, 0, 1], dtype=torch.uint8)
                e.reset(torch.tensor(init[::-1].copy()), mask=m)
                prev = None                                  # a masked reset rewrites the current observation in place
            obs, r, te, tr = e.step(torch.tensor(acts[k]))
            if prev is not None:
                np.testing.assert_array_equal(prev.cpu().numpy(), prev_copy)      # step k's observation survives step k+1
            prev, prev_copy = obs, obs.cpu().numpy().copy()
            res.append((obs.cpu().numpy().copy(), e.u.cpu().numpy().copy(), r.cpu().numpy().copy(), te.cpu().numpy().copy(),
                        tr.cpu().numpy().copy(), e.time_index.cpu().numpy().copy(), e.t["final_obs"].cpu().numpy().copy(),
                        e.t["beta"].cpu().numpy().copy()))
        outs.append(res)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    ended = 0
    for a, b in zip(outs[0][1:], outs[1][1:]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a[0], a[1])          # the observation IS the row
        ended += int(a[3].sum() + a[4].sum())
    assert ended > 0


def test_state_in_abi_validation():
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    env = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda")
    env.reset(torch.ones(2, 100), torch.ones(2, 100))
    assert env.state_in_obs and env.u is env.t["obs"]
    with pytest.raises(ValueError, match="out_obs"):
        env.step(torch.zeros(2), out_obs=env.t["obs"])
    env.params.sensing = N.SENSE_LAST                # state_in with a scalar observation: refused by the C side
    with pytest.raises(N.NativeError, match="state_in"):
        env.step(torch.zeros(2))
    env.params.sensing = N.SENSE_FULL
    env.step(torch.zeros(2))
    hist = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda", record_history=True)
    assert not hist.state_in_obs                      # the history mode keeps its own state rows


@pytest.mark.parametrize("kind,nx,S,burgers,B,T", [
    ("parabolic", 256, 100, False, 9, 11), ("transport", 100, 30, False, 9, 11), ("transport", 512, 20, True, 9, 11),
    ("parabolic", 40, 7, False, 9, 11), ("transport", 1500, 3, False, 9, 11),
    ("transport", 3, 2, False, 1, 5), ("parabolic", 2, 1, False, 130, 9), ("parabolic", 2047, 2, False, 3, 6), ("transport", 64, 5, False, 5, 1),
    # round 5: launches longer than one command batch (64 env-steps per load), S = 1, rows that fill the wave exactly (FULL kernels:
    # 64 / 128 / 256 / 512 slots) and rows that do not
    ("parabolic", 256, 1, False, 5, 130), ("transport", 128, 1, False, 3, 64), ("transport", 128, 3, False, 3, 65),
    ("parabolic", 100, 1, False, 4, 200), ("parabolic", 64, 2, False, 6, 129), ("transport", 512, 1, True, 2, 70)])
def test_rollout_kernel_equals_step_calls_bitwise(kind, nx, S, burgers, B, T):
    """pdegym_*_rollout (T env-steps in one launch, the row never leaves the wave's cache path) against T step calls: every
    observation slot, reward, flag, the time index, |u| sum, norm ring, restart counters, redrawn beta rows and the kept
    terminal observations agree bit for bit, across episode ends with the fused auto-reset."""
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    kw = dict(T=4 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=True, max_control_value=5.0, limit_pde_state_size=True, max_state_value=1e6)
    if burgers:
        kw["flux"] = "burgers"
    rng = np.random.default_rng(nx * 7 + S)
    envs = [PDEBatch1D(kind, reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B, device="cuda", **kw)
            for _ in range(2)]
    n = envs[0].n
    init = rng.uniform(0.5, 2, (B, n)).astype(np.float32)
    beta = rng.uniform(-2, 2, (B, n)).astype(np.float32)
    pool_i = rng.uniform(0.5, 2, (2 * B, n)).astype(np.float32)
    pool_b = rng.uniform(-2, 2, (2 * B, n)).astype(np.float32)
    acts = torch.tensor(rng.uniform(-1, 1, (T, B)).astype(np.float32), device="cuda")
    bufs = []
    for e in envs:
        assert e.can_rollout()
        e.reset(torch.tensor(init), torch.tensor(beta))
        e.enable_auto_reset(torch.tensor(pool_i), keep_final_obs=True, beta_pool=torch.tensor(pool_b))
        obs = torch.zeros(T + 1, B, n, device="cuda")
        obs[0].copy_(e.t["obs"])
        bufs.append((obs, torch.zeros(T, B, device="cuda"), torch.zeros(T, B, dtype=torch.uint8, device="cuda"),
                     torch.zeros(T, B, dtype=torch.uint8, device="cuda")))
    # (a) T step calls through the rollout buffers
    e, (obs, rew, te, tr) = envs[0], bufs[0]
    e.t["obs"] = obs[0]
    e.t["u"] = obs[0]
    for t in range(T):
        e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
    # (b) one rollout launch
    e2, (obs2, rew2, te2, tr2) = envs[1], bufs[1]
    e2.rollout(obs2, acts, rew2, te2, tr2)
    for a, b in ((obs, obs2), (rew, rew2), (te, te2), (tr, tr2)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    for k in ("time_index", "bsum", "ring", "reset_count", "beta", "final_obs", "norm_now", "norm_back"):
        np.testing.assert_array_equal(e.t[k].cpu().numpy(), e2.t[k].cpu().numpy(), err_msg=k)
    np.testing.assert_array_equal(e2.t["obs"].cpu().numpy(), obs2[T].cpu().numpy())
    assert T < 4 or int(te.sum() + tr.sum()) > 0
    # the engine carries on from slot T with ordinary step calls
    o_a = e.step(acts[0])[0].cpu().numpy()
    o_b = e2.step(acts[0])[0].cpu().numpy()
    np.testing.assert_array_equal(o_a, o_b)


@pytest.mark.parametrize("kind,nx,S,T", [("parabolic", 256, 1, 150), ("transport", 100, 2, 70), ("parabolic", 128, 3, 9)])
def test_rollout_without_auto_reset_runs_past_the_episode_end_like_step_calls(kind, nx, S, T):
    """Round 5 (the carried rollout stores time_index / bsum / norm_now / norm_back and the norm ring once, at the end of the
    launch): without reset pools an episode that ends inside the launch just stops advancing (nsub = 0), exactly as step calls do --
    every slot, flag and every word of engine state equals T step calls, also when the launch ends long after the episode."""
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    B = 5
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    kw = dict(T=37 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=True, max_control_value=5.0, limit_pde_state_size=True, max_state_value=1e6)
    rng = np.random.default_rng(nx + S)
    envs = [PDEBatch1D(kind, reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B, device="cuda", **kw)
            for _ in range(2)]
    n = envs[0].n
    init = rng.uniform(0.5, 2, (B, n)).astype(np.float32)
    beta = rng.uniform(-2, 2, (B, n)).astype(np.float32)
    acts = torch.tensor(rng.uniform(-1, 1, (T, B)).astype(np.float32), device="cuda")
    outs = []
    for mode, e in zip(("steps", "rollout"), envs):
        e.reset(torch.tensor(init), torch.tensor(beta))
        obs = torch.zeros(T + 1, B, n, device="cuda")
        obs[0].copy_(e.t["obs"])
        rew, te, tr = torch.zeros(T, B, device="cuda"), torch.zeros(T, B, dtype=torch.uint8, device="cuda"), torch.zeros(T, B, dtype=torch.uint8, device="cuda")
        if mode == "steps":
            e.t["obs"] = obs[0]
            e.t["u"] = obs[0]
            for t in range(T):
                e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
        else:
            e.rollout(obs, acts, rew, te, tr)
        outs.append([x.cpu().numpy().copy() for x in (obs, rew, te, tr)] + [e.t[k].cpu().numpy().copy() for k in
                                                                            ("time_index", "bsum", "ring", "norm_now", "norm_back")])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert (T < 37) == (outs[0][2].sum() == 0)              # the long launches end after the episode did


@pytest.mark.parametrize("kind,control,loc,stype,nx,S,B,T", [
    ("transport", "Neumann", "full", None, 100, 30, 9, 11), ("parabolic", "Neumann", "full", None, 256, 100, 9, 11),
    ("parabolic", "Neumann", "collocated", None, 64, 7, 9, 11), ("transport", "Dirchilet", "collocated", None, 100, 30, 9, 11),
    ("transport", "Dirchilet", "opposite", "Dirchilet", 512, 20, 5, 9), ("transport", "Neumann", "opposite", "Neumann", 40, 7, 130, 9),
    ("parabolic", "Dirchilet", "opposite", "Neumann", 1500, 3, 3, 9), ("parabolic", "Dirchilet", "collocated", None, 2, 1, 9, 9),
    ("transport", "Neumann", "full", None, 2047, 2, 3, 6), ("transport", "Neumann", "collocated", None, 3, 2, 1, 5)])
def test_general_rollout_kernel_equals_step_calls_bitwise(kind, control, loc, stype, nx, S, B, T):
    """Round 4: pdegym_*_rollout for the rest of the reference's control / sensing table (hyperbolic.py:66-124, parabolic.py:66-122):
    Neumann actuation (state through the observation slots) and scalar sensing (state in ``u``, the slots hold the sensed values).
    Every output and every piece of engine state equals T step calls bit for bit, across episode ends with the fused auto-reset."""
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    kw = dict(T=4 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=control, sensing_loc=loc,
              sensing_type=stype, normalize=True, max_control_value=5.0, limit_pde_state_size=True, max_state_value=1e6)
    rng = np.random.default_rng(nx * 11 + S)
    envs = [PDEBatch1D(kind, reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B, device="cuda", **kw)
            for _ in range(2)]
    n, od = envs[0].n, envs[0].obs_dim
    assert od == (n if loc == "full" else 1)
    init = rng.uniform(0.5, 2, (B, n)).astype(np.float32)
    beta = rng.uniform(-2, 2, (B, n)).astype(np.float32)
    pool_i = rng.uniform(0.5, 2, (2 * B, n)).astype(np.float32)
    pool_b = rng.uniform(-2, 2, (2 * B, n)).astype(np.float32)
    acts = torch.tensor(rng.uniform(-1, 1, (T, B)).astype(np.float32), device="cuda")
    bufs = []
    for e in envs:
        assert e.can_rollout()
        e.reset(torch.tensor(init), torch.tensor(beta))
        e.enable_auto_reset(torch.tensor(pool_i), keep_final_obs=True, beta_pool=torch.tensor(pool_b))
        obs = torch.zeros(T + 1, B, od, device="cuda")
        obs[0].copy_(e.t["obs"])
        bufs.append((obs, torch.zeros(T, B, device="cuda"), torch.zeros(T, B, dtype=torch.uint8, device="cuda"),
                     torch.zeros(T, B, dtype=torch.uint8, device="cuda")))
    e, (obs, rew, te, tr) = envs[0], bufs[0]
    if e.state_in_obs:
        e.t["obs"] = obs[0]
        e.t["u"] = obs[0]
    for t in range(T):
        e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
    e2, (obs2, rew2, te2, tr2) = envs[1], bufs[1]
    e2.rollout(obs2, acts, rew2, te2, tr2)
    for a, b in ((obs, obs2), (rew, rew2), (te, te2), (tr, tr2)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    for k in ("time_index", "bsum", "ring", "reset_count", "beta", "final_obs", "norm_now", "norm_back", "u"):
        np.testing.assert_array_equal(e.t[k].cpu().numpy(), e2.t[k].cpu().numpy(), err_msg=k)
    np.testing.assert_array_equal(e2.t["obs"].cpu().numpy(), obs2[T].cpu().numpy())
    assert T < 4 or int(te.sum() + tr.sum()) > 0
    np.testing.assert_array_equal(e.step(acts[0])[0].cpu().numpy(), e2.step(acts[0])[0].cpu().numpy())


def test_rollout_abi_validation():
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    env = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda", sensing_loc="full", sensing_type=None, record_history=True)
    assert not env.can_rollout()            # a recorded history has no place in the rollout kernels
    with pytest.raises(ValueError):
        env.rollout(None, None, None, None, None)
    env = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda", sensing_loc="opposite", sensing_type="Dirchilet")
    assert env.can_rollout() and env.obs_dim == 1
    env = PDEBatch1D("transport", 1, 1e-3, 1, 1e-2, 0.01, num_envs=2, device="cuda", sensing_loc="full", sensing_type=None)
    env.reset(torch.ones(2, 100), torch.ones(2, 100))
    n = env.n
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")   # noqa: E731
    with pytest.raises(Exception, match="obs"):
        env.rollout(z(4, 2, n), z(4, 2), z(4, 2), z(4, 2, dt=torch.uint8), z(4, 2, dt=torch.uint8))   # T + 1 slots needed


@pytest.mark.parametrize("kind,nx,S,dx_pow2", [
    ("transport", 100, 7, False),     # 50 full lanes of two slots, 14 empty: the common loop (no straddling lane)
    ("transport", 101, 3, False),     # the last lane straddles the row's end: the loop with the slot-by-slot tail
    ("transport", 64, 5, True),       # dx = 2^-6: the exact-multiply quotient, one slot per lane, FULL-sized row
    ("transport", 130, 4, False),     # EPL = 3: vector store of 2 + 1
    ("transport", 513, 2, False),     # EPL = 9 -> instantiated as 12: no fast history loop (the select form serves it)
    ("parabolic", 200, 9, False),     # the reference's example grid: 200 slots = 50 full lanes of four
    ("parabolic", 201, 3, False),     # straddling lane; node 0 of every stored row must read 0
    ("parabolic", 256, 6, False),     # the C2 row
])
def test_history_mode_fast_loop_writes_the_reference_trajectory(kind, nx, S, dx_pow2):
    """HFAST (round 6: fast arithmetic + one trajectory-row store per sub-step, what a single environment with record_history runs):
    every row of ``history`` equals the oracle's kept trajectory bit for bit -- through an episode end (fewer sub-steps than S in the
    last call, then post-terminal calls that write nothing), for an instance whose step is redone by the exact loop (a state that
    overflows float32 inside the call), for a commanded boundary value of exactly -0.0, and next to ordinary instances in the same
    launch."""
    from oracle import pde_oracle as po
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    n = nx + (kind == "parabolic")
    dx = 2.0 ** -6 if dx_pow2 else 1.0 / nx
    dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
    steps = 6
    nt1 = S * steps - 2                                  # the last call has S - 2 sub-steps left
    kw = dict(T=nt1 * dt, dt=dt, X=nx * dx, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=False, max_control_value=20, limit_pde_state_size=True, max_state_value=1e10)
    B = 5
    rng = np.random.default_rng(nx + S)
    init = rng.uniform(-2, 2, (B, n)).astype(np.float32)
    # overflows float32 inside the call (parabolic: 2*u of parabolic.py:143 at |u| >= 2^127): non-finite rows, the exact redo
    init[3] *= np.float32(1.6e38 if kind == "parabolic" else 3e37)
    init[4, ::3] = 0.0
    beta = rng.uniform(-5, 5, (B, n)).astype(np.float32)
    # (no reward: with fewer than 100 rows the reference's look-back index u[t - 100] is out of range, tuned_reward_1d.py:40)
    orc = (po.ParabolicOracle if kind == "parabolic" else po.TransportOracle)(reward=None, keep_history=True, **kw)
    env = PDEBatch1D(kind, reward=RewardSpec(N.REWARD_NONE), num_envs=B, device="cuda", record_history=True, **kw)
    orc.reset(init, beta)
    env.reset(torch.tensor(init), torch.tensor(beta))
    with np.errstate(all="ignore"):
        for k in range(steps + 2):
            a = rng.uniform(-3, 3, B).astype(np.float32)
            a[1] = np.float32(-0.0)                      # x + 0*t would turn a frozen -0.0 into +0.0: the exact loop takes it
            o_ref, r_ref, te_ref, tr_ref = orc.step(a)
            o, r, te, tr = env.step(torch.tensor(a))
            np.testing.assert_array_equal(o.cpu().numpy().view(np.uint32), np.asarray(o_ref, dtype=np.float32).view(np.uint32), err_msg=f"step {k}")
            np.testing.assert_array_equal(te.cpu().numpy().astype(bool), te_ref)
            np.testing.assert_array_equal(env.time_index.cpu().numpy(), orc.time_index)
    hg, ho = env.t["history"].cpu().numpy(), orc.hist
    assert hg.shape == ho.shape == (B, nt1 + 1, n)
    np.testing.assert_array_equal(hg.view(np.uint32), ho.view(np.uint32))
    assert te_ref.all() and not np.isfinite(ho[3]).all() and np.isfinite(ho[0]).all()
    if kind == "parabolic":
        assert (hg[:, 1:, 0] == 0).all()
