"""CPU tests: the C-ABI library loads and exports every symbol of include/pdegym.h, and the host-side mirror
of the reference interface behaves like the reference (names, spaces, errors, return types, auto-reset).

Compute in these tests goes through tests/fake_backend.py (the oracle on CPU tensors); no kernel is launched.
"""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from tests.cases import NS_BC, PARABOLIC_CASES, TRANSPORT_CASES
from tests.fake_backend import FakeBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ------------------------------------------------------------------------------------------
def test_library_builds_and_exports_every_declared_symbol():
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd import build
    lib = build.build()
    header = open(os.path.join(ROOT, "include", "pdegym.h")).read()
    declared = set(re.findall(r"\b(pdegym_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.EXPORTS), declared ^ set(N.EXPORTS)
    h = ctypes.CDLL(lib)
    for sym in declared:
        assert hasattr(h, sym), sym
    assert N.load().pdegym_abi_version() == N.ABI_VERSION


def test_ctypes_structs_match_header_layout():
    """Field order/names of the ctypes mirrors follow the C structs (a reordering would silently corrupt calls)."""
    from pdecontrolgym_amd import _native as N
    header = open(os.path.join(ROOT, "include", "pdegym.h")).read()

    def names(decls):
        out = []
        for d in decls:
            out.extend(x.strip().lstrip("*").split("[")[0] for x in d.split(","))
        return out

    for struct, mirror in (("pdegym_params1d", N.Params1D), ("pdegym_bufs1d", N.Bufs1D), ("pdegym_rollout1d", N.Rollout1D),
                           ("pdegym_params_ns2d", N.ParamsNS2D), ("pdegym_bufs_ns2d", N.BufsNS2D),
                           ("pdegym_rollout_traffic", N.RolloutTraffic), ("pdegym_bufs_traffic", N.BufsTraffic)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        decls = []
        for d in body.split(";"):
            d = d.strip()
            if not d:
                continue
            d = re.sub(r"^(const\s+)?(struct\s+\w+|int32_t|int64_t|uint8_t|float|double|void)\s*\*?", "", d).strip()
            decls.append(d)
        assert names(decls) == [f[0] for f in mirror._fields_], struct


def test_native_refuses_cpu_tensors():
    import torch
    from pdecontrolgym_amd import _native as N
    with pytest.raises(N.NativeError):
        N.dptr(torch.zeros(4))


def test_no_product_module_imports_the_oracle():
    for pkg in ("pdecontrolgym_amd", "pde_control_gym"):
        for dp, _, fs in os.walk(os.path.join(ROOT, pkg)):
            for f in fs:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dp, f)


# ---- import surface (reference pde_control_gym/__init__.py, src/__init__.py) ----------------------------
def test_import_surface_and_registration():
    import pde_control_gym
    from pde_control_gym.src import (BaseReward, NavierStokes2D, NormReward, NSReward, ReactionDiffusionPDE1D,  # noqa: F401
                                     TrafficARZReward, TrafficPDE1D, TransportPDE1D, TunedReward1D)
    from pde_control_gym.src.environments2d.navier_stokes2D import central_difference, laplace
    for i in ("PDEControlGym-TransportPDE1D", "PDEControlGym-ReactionDiffusionPDE1D", "PDEControlGym-NavierStokes2D",
              "PDEControlGym-BrainTumor1D", "PDEControlGym-TrafficPDE1D"):
        assert i in pde_control_gym._IDS
    from pde_control_gym.src import BrainTumor1D, BrainTumorReward, TherapyWrapper  # noqa: F401
    from pde_control_gym.src.environments1d.brain_tumor_env import TherapyWrapper as W2
    assert W2 is TherapyWrapper
    f = np.arange(25.0).reshape(5, 5) ** 2
    d = central_difference(f, "x", 0.5)
    assert d[0].sum() == 0 and d[2, 2] == (f[2, 3] - f[2, 1]) / 1.0
    assert laplace(f, 1.0, 1.0)[2, 2] == f[2, 1] + f[1, 2] - 4 * f[2, 2] + f[2, 3] + f[3, 2]


def _transport_params(**over):
    from pde_control_gym.src import TunedReward1D
    T, dt = 1, 1e-4
    nx = 100
    beta = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, nx)))).astype(np.float32)
    p = {"T": T, "dt": dt, "X": 1, "dx": 1e-2, "reward_class": TunedReward1D(int(round(T / dt)), -1e3, 3e2),
         "normalize": False, "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None,
         "sensing_noise_func": lambda state: state, "limit_pde_state_size": True, "max_state_value": 1e10,
         "max_control_value": 20, "reset_init_condition_func": lambda nx: np.ones(nx) * 5.0,
         "reset_recirculation_func": lambda nx: beta, "control_sample_rate": 0.1}
    p.update(over)
    return p


def test_make_transport_matches_reference_golden_through_the_public_api(golden_transport):
    """gym.make-style construction from the example's parameter dict (transport1Dppo.py:59-77), NumPy API."""
    import pde_control_gym
    g = golden_transport["H1"]
    env = pde_control_gym.make("PDEControlGym-TransportPDE1D", device="cpu", backend=FakeBackend(), **_transport_params())
    assert env.nt == 10001 and env.nx == 100
    assert env.action_space.shape == (1,) and env.action_space.dtype == np.float32
    assert env.observation_space.shape == (100,) and env.observation_space.high[0] == np.float32(1e10)
    obs, info = env.reset(seed=3, options={"x": 1})
    assert info == {} and obs.dtype == np.float32 and obs.shape == (100,)
    np.testing.assert_array_equal(obs, g.obs[0])
    for i, a in enumerate(g.actions):
        obs, r, te, tr, info = env.step(np.array([a], dtype=np.float32))     # SB3 passes a (1,) float32 array
        assert isinstance(te, bool) and isinstance(tr, bool) and info == {}
        np.testing.assert_array_equal(obs, g.obs[i + 1])
        np.testing.assert_allclose(r, g.reward[i], rtol=1e-6, atol=1e-4)
        assert env.time_index == g.time_index[i] and env.terminate() == bool(g.terminate[i])
    assert te and env.u.shape == (10001, 100)
    np.testing.assert_array_equal(env.u[env.time_index], g.rows[-1])
    # python float and 0-d actions are accepted too
    env.reset()
    o1 = env.step(0.25)[0]
    env.reset()
    o2 = env.step(np.float32(0.25))[0]
    np.testing.assert_array_equal(o1, o2)


def test_error_conventions_match_reference():
    from pde_control_gym.src import ReactionDiffusionPDE1D, TransportPDE1D
    fb = dict(device="cpu", backend=FakeBackend())
    with pytest.raises(Exception, match="Invalid sensing_loc parameter"):
        TransportPDE1D(**_transport_params(sensing_loc="nope"), **fb)
    with pytest.raises(Exception, match="Invalid control_type parameter"):
        TransportPDE1D(**_transport_params(control_type="Dirichlet"), **fb)      # correct spelling is NOT accepted
    with pytest.raises(Exception, match="Invalid sensing_type parameter"):
        TransportPDE1D(**_transport_params(sensing_loc="opposite", sensing_type="x"), **fb)
    with pytest.raises(Exception, match="Dirchilet sensing at u\\(0, t\\) is not viable"):
        ReactionDiffusionPDE1D(**_transport_params(sensing_loc="opposite", sensing_type="Dirchilet"), **fb)
    env = TransportPDE1D(**_transport_params(reset_init_condition_func=lambda nx: 1 / 0), **fb)
    with pytest.raises(Exception, match="Please pass both an initial condition and a recirculation function"):
        env.reset()
    from pde_control_gym.src import NavierStokes2D, NSReward
    with pytest.raises(RuntimeError, match="Stability is not guarenteed"):
        NavierStokes2D(T=0.2, dt=1e-2, X=1, dx=0.05, Y=1, dy=0.05, action_dim=1, reward_class=NSReward(0.1),
                       normalize=False, reset_init_condition_func=None, boundary_condition=NS_BC,
                       U_ref=np.zeros((20, 21, 21, 2)), action_ref=np.ones(20), **fb)


def test_sensing_modes_and_scalar_observations(golden_transport):
    from pde_control_gym.src import TransportPDE1D
    g = golden_transport["H2_dir_col"]
    kw = dict(TRANSPORT_CASES["H2_dir_col"])
    p = _transport_params(**kw)
    p["reset_init_condition_func"] = lambda nx: g.init
    p["reset_recirculation_func"] = lambda nx: g.beta
    from pde_control_gym.src import TunedReward1D
    p["reward_class"] = TunedReward1D(3000, -1e3, 3e2)
    env = TransportPDE1D(device="cpu", backend=FakeBackend(), **p)
    assert env.observation_space.shape == (1,)
    obs, _ = env.reset()
    assert np.ndim(obs) == 0 and obs == g.obs[0][0]
    for i, a in enumerate(g.actions):
        obs, r, te, tr, _ = env.step(np.array([a], dtype=np.float32))
        assert obs == g.obs[i + 1][0]


# the export / custom-reward / NS public-API tests run on the CPU double here and on the real HIP backend under -m gpu
BACKENDS = [pytest.param("double", id="cpu-double"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _bk(kind):
    return dict(device="cpu", backend=FakeBackend()) if kind == "double" else dict(device="cuda")


@pytest.mark.parametrize("bk", BACKENDS)
def test_export_1d_trajectory(golden_transport, tmp_path, bk):
    import pde_control_gym
    from pde_control_gym import export
    g = golden_transport["H1"]
    env = pde_control_gym.make("PDEControlGym-TransportPDE1D", record_history=True, **_bk(bk),
                               **_transport_params()).unwrapped
    env.reset()
    acts, rews = [], []
    for a in g.actions[:3]:
        _, r, *_ = env.step(np.array([a], dtype=np.float32))
        acts.append(a)
        rews.append(r)
    export.save_trajectory_1d(tmp_path / "traj.npz", env, acts, rews)
    z = np.load(tmp_path / "traj.npz")
    assert z["u"].shape == (env.nt, 100) and int(z["time_index"]) == env.time_index and len(z["rewards"]) == 3
    np.testing.assert_array_equal(z["u"][env.time_index], g.rows[2])
    env2 = pde_control_gym.make("PDEControlGym-TransportPDE1D", record_history=False, **_bk(bk),
                                **_transport_params()).unwrapped
    env2.reset()
    with pytest.raises(ValueError, match="record_history"):
        export.save_trajectory_1d(tmp_path / "no.npz", env2)


@pytest.mark.parametrize("bk", BACKENDS)
def test_custom_reward_class_gets_a_trajectory_view(golden_transport, bk):
    """A user BaseReward subclass (docs/source/utils/customrewards.rst) is called with (uVec, t, term, trunc, action)."""
    from pde_control_gym.src import BaseReward, TransportPDE1D
    g = golden_transport["H1"]
    calls = []

    class MyReward(BaseReward):
        def reward(self, uVec=None, time_index=None, terminate=None, truncate=None, action=None):
            calls.append((time_index, terminate, truncate, float(action)))
            return float(-np.linalg.norm(uVec[time_index]) + uVec[0][0] + np.abs(uVec[:, -1]).sum() * 0)

    env = TransportPDE1D(**_bk(bk), **_transport_params(reward_class=MyReward()))
    env.reset()
    for i, a in enumerate(g.actions[:3]):
        obs, r, te, tr, _ = env.step(np.float32(a))
        np.testing.assert_allclose(r, -np.linalg.norm(g.rows[i]) + 5.0, rtol=1e-6)
    assert calls[0][0] == 1000 and calls[0][3] == pytest.approx(float(g.actions[0]))


@pytest.mark.parametrize("bk", BACKENDS)
def test_vecenv_with_a_separate_state_tolerates_in_place_edits_of_the_observation(bk):
    """make_vec(..., state_in_obs=False): the observation tensor is an output only, so a caller that scales it in place does not
    touch the plant -- trajectories equal those of the default engine (observation = state, read-only) step for step."""
    import itertools
    import torch
    import pde_control_gym
    B = 3
    ics = [np.linspace(1.0, 2.0 + k, 100).astype(np.float32) for k in range(B)]
    mk = lambda **extra: pde_control_gym.make_vec(  # noqa: E731
        "PDEControlGym-TransportPDE1D", num_envs=B, **_bk(bk),
        **_transport_params(T=0.05, dt=1e-4, control_sample_rate=0.01, reset_init_condition_func=(lambda it: lambda nx: next(it))(itertools.cycle(ics)), **extra))
    ref, sep = mk(), mk(state_in_obs=False)
    assert ref.core.state_in_obs and not sep.core.state_in_obs
    o_ref, o_sep = ref.reset_tensor(), sep.reset_tensor()
    assert torch.equal(o_ref, o_sep)
    rng = np.random.default_rng(3)
    for i in range(4):
        a = torch.as_tensor(rng.uniform(-1, 1, (B, 1)).astype(np.float32), device=o_ref.device)
        o_sep.mul_(0.0)                              # in-place post-processing of the previous observation
        o_ref, r_ref, d_ref, _ = ref.step_tensor(a)
        o_sep, r_sep, d_sep, _ = sep.step_tensor(a)
        assert torch.equal(o_ref, o_sep) and torch.equal(r_ref, r_sep) and torch.equal(d_ref, d_sep)


@pytest.mark.parametrize("bk", BACKENDS)
@pytest.mark.parametrize("norm", ["1", "2", "inf"])
def test_norm_reward_differential_is_evaluated_by_the_step_kernel(bk, norm):
    """NormReward(horizon="differential") maps onto the in-kernel reward (reward_horizon of pdegym_params1d; no host callback,
    no recorded trajectory on the batched face) and equals the host definition ||u[t] - u[t-1]|| over fine-time rows, evaluated
    on the trajectory of a single environment that records it; "t-horizon" maps onto PDEGYM_HORIZON_T (test below)."""
    import pde_control_gym
    from pde_control_gym.src import NormReward, TransportPDE1D
    from pde_control_gym.src.environments1d.base_env_1d import reward_spec_for
    from pdecontrolgym_amd import _native as N
    assert reward_spec_for(NormReward(10, norm, "differential")).horizon == N.HORIZON_DIFFERENTIAL
    assert reward_spec_for(NormReward(10, norm, "temporal")).horizon == N.HORIZON_TEMPORAL
    th = reward_spec_for(NormReward(10, norm, "t-horizon", t_horizon_length=7))
    assert th.horizon == N.HORIZON_T and th.t_horizon == 7
    assert reward_spec_for(NormReward(10, norm, "t-horizon", t_horizon_length=129)) is None       # beyond the ring of row norms: host path
    B, T, dt = 3, 0.05, 1e-4
    ics = [np.linspace(1.0, 2.0 + k, 100).astype(np.float32) for k in range(B)]
    import itertools
    it = itertools.cycle(ics)                 # the episode-end auto reset draws initial conditions again
    rw = NormReward(int(round(T / dt)), norm, "differential", -2.0, 55.0)
    p = _transport_params(T=T, dt=dt, control_sample_rate=0.01, reward_class=rw, reset_init_condition_func=lambda nx: next(it))
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_bk(bk), **p)
    assert not venv._host_reward and venv.core.t["history"] is None and not venv.core.can_rollout()
    venv.reset()
    singles = []
    for b in range(B):
        e = TransportPDE1D(**_bk(bk), **_transport_params(T=T, dt=dt, control_sample_rate=0.01, reward_class=rw,
                                                          reset_init_condition_func=lambda nx, b=b: ics[b]))
        e.reset()
        singles.append(e)
    rng = np.random.default_rng(5)
    for i in range(5):
        a = rng.uniform(-1, 1, (B, 1)).astype(np.float32)
        _, r, dones, _ = venv.step(a)
        for b, e in enumerate(singles):
            _, r1, te, tr, _ = e.step(a[b])
            want = rw.reward(e.u, e.time_index, te, tr, a[b])
            assert r1 == pytest.approx(want, rel=1e-5, abs=1e-6) and r[b] == pytest.approx(want, rel=1e-5, abs=1e-6)
            assert (want == 55.0) == te and (te or want > 0)
    assert dones.all()


@pytest.mark.parametrize("bk", BACKENDS)
@pytest.mark.parametrize("norm,klen", [("1", 3), ("2", 150), ("inf", 30)])
def test_norm_reward_t_horizon_is_evaluated_by_the_step_kernel(bk, norm, klen):
    """NormReward(horizon="t-horizon") -- host-only until round 4 -- maps onto PDEGYM_HORIZON_T: -(mean of the norms of the last
    min(t_horizon_length, t + 1) fine-time rows), no recorded trajectory on the batched face; equals the host definition evaluated
    on the trajectory of single environments (100 sub-steps per env-step: lengths inside one step; 150 > the ring of 128 norms takes
    the host path on the recorded trajectory, same values)."""
    import itertools
    import pde_control_gym
    from pde_control_gym.src import NormReward, TransportPDE1D
    B, T, dt = 3, 0.05, 1e-4
    ics = [np.linspace(1.0, 2.0 + k, 100).astype(np.float32) for k in range(B)]
    it = itertools.cycle(ics)
    rw = NormReward(int(round(T / dt)), norm, "t-horizon", -2.0, 55.0, klen)
    p = _transport_params(T=T, dt=dt, control_sample_rate=0.01, reward_class=rw, reset_init_condition_func=lambda nx: next(it))
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_bk(bk), **p)
    in_kernel = klen <= 128
    assert venv._host_reward == (not in_kernel) and (venv.core.t["history"] is None) == in_kernel and not venv.core.can_rollout()
    venv.reset()
    singles = []
    for b in range(B):
        e = TransportPDE1D(**_bk(bk), **_transport_params(T=T, dt=dt, control_sample_rate=0.01, reward_class=rw,
                                                          reset_init_condition_func=lambda nx, b=b: ics[b]))
        e.reset()
        singles.append(e)
    rng = np.random.default_rng(6)
    for i in range(5):
        a = rng.uniform(-1, 1, (B, 1)).astype(np.float32)
        _, r, dones, _ = venv.step(a)
        for b, e in enumerate(singles):
            _, r1, te, tr, _ = e.step(a[b])
            want = rw.reward(e.u, e.time_index, te, tr, a[b])
            assert r1 == pytest.approx(want, rel=1e-5, abs=1e-6) and r[b] == pytest.approx(want, rel=1e-5, abs=1e-6)
            assert (want == 55.0) == te and (te or want < 0)
    assert dones.all()


@pytest.mark.parametrize("bk", BACKENDS)
def test_vecenv_accepts_custom_reward_classes(golden_transport, bk):
    """docs/source/utils/customrewards.rst on the batched face: a user BaseReward subclass is evaluated per instance on a
    lazy view of the device-resident trajectory (slow compatibility path); values equal those of single environments."""
    import pde_control_gym
    from pde_control_gym.src import BaseReward, TransportPDE1D
    g = golden_transport["H1"]

    class MyReward(BaseReward):
        def reward(self, uVec=None, time_index=None, terminate=None, truncate=None, action=None):
            return float(-np.linalg.norm(uVec[time_index]) + 0.5 * np.linalg.norm(uVec[time_index - 100]) + 0.1 * float(action) + 7 * terminate)

    B = 3
    ics = [np.ones(100, dtype=np.float32) * (2.0 + k) for k in range(B)]
    it = iter(ics)
    p = _transport_params(reward_class=MyReward(), reset_init_condition_func=lambda nx: next(it))
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_bk(bk), **p)
    venv.reset()
    singles = []
    for b in range(B):
        e = TransportPDE1D(**_bk(bk), **_transport_params(reward_class=MyReward(), reset_init_condition_func=lambda nx, b=b: ics[b]))
        e.reset()
        singles.append(e)
    for i in range(3):
        a = np.array([[0.3], [-0.7], [0.1]], dtype=np.float32) * (i + 1)
        obs, rew, dones, infos = venv.step(a)
        for b in range(B):
            o1, r1, te, tr, _ = singles[b].step(a[b])
            np.testing.assert_array_equal(obs[b], o1)
            np.testing.assert_allclose(rew[b], r1, rtol=1e-6)
    with pytest.raises(NotImplementedError):
        venv.enable_fused_auto_reset()


@pytest.mark.parametrize("bk", BACKENDS)
def test_vecenv_custom_reward_navier_stokes(golden_ns, bk):
    """The reference's extension point is environment-agnostic (docs/source/utils/customrewards.rst:8-9, base_reward.py:13-32):
    a user BaseReward subclass on the batched NavierStokes2D face is called as navier_stokes2D.py:151 calls it, per instance,
    on a lazy view of the device-resident trajectory.  A re-implementation of NSReward must reproduce the in-kernel reward, a
    reward that looks one frame back must equal the same class on single environments."""
    import pde_control_gym
    from pde_control_gym.src import BaseReward, NavierStokes2D, NSReward
    g = golden_ns["N1"]

    class Again(BaseReward):          # NSReward restated: must agree with the kernel's epilogue
        def reward(self, uVec=None, time_index=None, U_ref=None, action=None, action_ref=None):
            d = np.asarray(uVec[time_index]) - U_ref[time_index]
            return -0.5 * np.linalg.norm(d) ** 2 / uVec.shape[1] / uVec.shape[2] - 0.05 * np.linalg.norm(action - action_ref[time_index]) ** 2

    class Back(BaseReward):           # needs an earlier frame: the trajectory view, not only the current observation
        def reward(self, uVec=None, time_index=None, U_ref=None, action=None, action_ref=None):
            return float(np.abs(np.asarray(uVec[time_index]) - np.asarray(uVec[time_index - 1])).sum() + 0.25 * float(np.sum(action)))

    B = 3
    scale = [1.0, 0.5, -0.75]
    Uref = np.full((200, 21, 21, 2), 0.1)

    def params(rc, k=None):
        it = iter(range(B))
        return {"T": 0.2, "dt": 1e-3, "X": 1, "dx": 0.05, "Y": 1, "dy": 0.05, "action_dim": 1, "reward_class": rc, "normalize": False,
                "reset_init_condition_func": (lambda X: (g.u0 * scale[next(it)], g.v0.copy(), np.zeros_like(X))) if k is None
                else (lambda X: (g.u0 * scale[k], g.v0.copy(), np.zeros_like(X))),
                "boundary_condition": NS_BC, "U_ref": Uref, "action_ref": 2.0 * np.ones(1000), "maximum_pressure_iteration": 30}

    acts = [np.array([[3.5], [2.5], [3.0]]), np.array([[2.2], [3.9], [2.0]]), np.array([[3.1], [3.1], [2.6]])]
    venv_k = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, dtype="float64", **_bk(bk), **params(NSReward(0.1)))
    venv_a = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, dtype="float64", **_bk(bk), **params(Again()))
    venv_b = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, dtype="float64", **_bk(bk), **params(Back()))
    singles = [NavierStokes2D(**_bk(bk), **params(Back(), k)) for k in range(B)]
    for v in (venv_k, venv_a, venv_b):
        v.reset()
    for e in singles:
        e.reset()
    for a in acts:
        ok, rk, *_ = venv_k.step(a)
        oa, ra, *_ = venv_a.step(a)
        ob, rb, *_ = venv_b.step(a)
        np.testing.assert_array_equal(ok, oa)
        np.testing.assert_allclose(ra, rk, rtol=1e-6)          # (the SB3 face hands rewards out in float32)
        for k in range(B):
            o1, r1, *_ = singles[k].step(a[k])
            np.testing.assert_array_equal(ob[k], o1.astype(np.float32))
            np.testing.assert_allclose(rb[k], r1, rtol=1e-6)
    with pytest.raises(NotImplementedError):
        venv_b.enable_fused_auto_reset()


def test_device_sensing_noise_hook_on_the_cpu_double():
    """sensing_noise_tensor_func on the torch-native face and in the eager DeviceRollout: the policy sees f(obs), the state stays
    clean, the trajectory is that of the policy composed with f (the graph-captured form runs under -m gpu)."""
    import torch
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    f = lambda o: o * 1.5 - 0.125
    pol = lambda o: torch.tanh(o.mean(dim=1))

    def venv(noise):
        v = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=3, device="cpu", backend=FakeBackend(),
                                     sensing_noise_tensor_func=noise, **_transport_params(reset_init_condition_func=lambda nx: np.ones(nx) * 2.0))
        v.reset_tensor()
        return v

    a = DeviceRollout(venv(f), pol, 4, use_graph=False).run()
    b = DeviceRollout(venv(None), lambda o: pol(f(o)), 4, use_graph=False).run()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.actions, b.actions) and torch.equal(a.obs_seen, f(a.obs)) and b.obs_seen is None
    v = venv(f)
    o, *_ = v.step_tensor(torch.zeros(3))
    assert torch.equal(o, f(v.core.t["obs"]))
    with pytest.raises(ValueError):
        DeviceRollout(venv(f), pol, 4, use_graph=False, one_launch=True)


def test_every_c_abi_call_runs_with_its_tensors_device_current(monkeypatch):
    """One process may drive several GPUs (one engine per device): every backend method that reaches the C ABI switches to the
    device of its tensors first (backend._on_device_of).  The guard cannot run for real on a one-GPU box (the -m gpu test needs
    two devices), so it is unit-tested here: entered with the tensors' device, for the dict form and the tensor form, not at
    all for CPU tensors -- and no public backend method lacks it.  (The multi-device path itself is hardware-unverified.)"""
    import torch
    from pdecontrolgym_amd import backend as bk_mod
    entered = []

    class Guard:
        def __init__(self, dev):
            entered.append(dev)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "device", Guard)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)

    class Stub:                        # stands in for a tensor on cuda:1 in the dict form
        device = torch.device("cuda", 1)

    class StubCurrent:                 # ... and for one on the device that is current already: no switch, no context manager
        device = torch.device("cuda", 0)

    class B:
        @bk_mod._on_device_of("obs")
        def by_dict(self, P, T, n):
            return ("ran", n)

        @bk_mod._on_device_of("u")
        def by_tensor(self, rows, out):
            return "ran"

    assert B().by_dict(None, {"obs": Stub()}, 3) == ("ran", 3) and entered == [torch.device("cuda", 1)]
    entered.clear()
    assert B().by_dict(None, {"obs": StubCurrent()}, 4) == ("ran", 4) and entered == []
    assert B().by_dict(None, {"obs": torch.zeros(1)}, 3) == ("ran", 3) and entered == []        # CPU tensors: no device switch
    assert B().by_tensor(torch.zeros(2), torch.zeros(2)) == "ran" and entered == []
    public = [n for n, f in vars(bk_mod.HipBackend).items() if callable(f) and not n.startswith("_") and n not in ("bind",)]
    assert len(public) >= 14
    for n in public:
        assert hasattr(getattr(bk_mod.HipBackend, n), "device_guard_key"), f"HipBackend.{n} reaches the C ABI without the device guard"


def test_golden_fixtures_are_exactly_what_the_generator_writes():
    """'Pinned' stays literally true: tests/golden/make_golden.py --check re-runs the reference (build container only) and
    compares every key of every committed .npz with what it would write today, bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.environ.get("PDEGYM_REFERENCE", "/root/reference")):
        pytest.skip("the reference checkout is not on this machine (GPU box): fixtures cannot be regenerated here")
    # as its docstring documents it: no PYTHONPATH, any working directory (the script puts the repository root on sys.path itself)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py"), "--check"], cwd=os.path.dirname(root),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    # importable, with its generator table at module level (advisor finding r3): check() of one small file from this process
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_mod", os.path.join(root, "tests", "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "gymnasium" or k.startswith(("gymnasium.", "pde_control_gym"))}
    try:
        spec.loader.exec_module(mod)
        assert set(mod.GEN) == set(mod.FILES)
    finally:
        for k in [k for k in sys.modules if k == "gymnasium" or k.startswith(("gymnasium.", "pde_control_gym"))]:
            if k not in saved:
                del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})


def test_parabolic_single_env_public_api(golden_parabolic):
    from pde_control_gym.src import ReactionDiffusionPDE1D, TunedReward1D
    g = golden_parabolic["P2_s100"]
    kw = dict(PARABOLIC_CASES["P2_s100"])
    env = ReactionDiffusionPDE1D(device="cpu", backend=FakeBackend(), reward_class=TunedReward1D(1000, -1e3, 3e2),
                                 sensing_noise_func=lambda s: s, reset_init_condition_func=lambda nx: g.init,
                                 reset_recirculation_func=lambda nx: g.beta, **kw)
    assert env.observation_space.shape == (257,) and env.nx == 256
    obs, _ = env.reset()
    for i, a in enumerate(g.actions):
        obs, r, te, tr, _ = env.step(np.array([a], dtype=np.float32))
        np.testing.assert_array_equal(obs, g.obs[i + 1])
        np.testing.assert_allclose(r, g.reward[i], rtol=1e-6, atol=1e-4)
        assert te == bool(g.terminate[i])


@pytest.mark.parametrize("bk", BACKENDS)
def test_ns_single_env_public_api_reproduces_target_frames(golden_ns, bk):
    """NS2Dppo.py:36-50 parameter dict; env.U / env.u / env.solve_pressure as used by NS2Doptimization.py."""
    from pde_control_gym.src import NavierStokes2D, NSReward
    g = golden_ns["N1"]
    Uref = np.zeros((200, 21, 21, 2))
    for t in (1, 2):                      # the golden rewards were produced with U_ref = the target trajectory itself
        Uref[t] = np.stack([g[f"u{t}"], g[f"v{t}"]], -1)
    p = {"T": 0.2, "dt": 1e-3, "X": 1, "dx": 0.05, "Y": 1, "dy": 0.05, "action_dim": 1, "reward_class": NSReward(0.1),
         "normalize": False, "reset_init_condition_func": lambda X: (g.u0.copy(), g.v0.copy(), np.zeros_like(X)),
         "boundary_condition": NS_BC, "U_ref": Uref, "action_ref": 2.0 * np.ones(1000), "maximum_pressure_iteration": 2000}
    env = NavierStokes2D(**_bk(bk), **p)
    assert (env.nt, env.nx, env.ny) == (200, 21, 21) and env.X.shape == (21, 21)
    assert env.observation_space.shape == (21, 21, 2) and env.action_space.shape == (1,)
    obs, info = env.reset(seed=400)
    assert obs.shape == (21, 21, 2) and obs.dtype == np.float64 and info == {}
    for t in (1, 2):
        obs, r, te, tr, info = env.step(g.actions[t - 1])
        np.testing.assert_array_equal(obs[..., 0], g[f"u{t}"])
        np.testing.assert_array_equal(env.U[t, :, :, 1], g[f"v{t}"])
        np.testing.assert_allclose(r, g.rewards[t - 1], rtol=1e-12)
        assert tr is False and te is False
    np.testing.assert_array_equal(env.u, g["u2"])
    pr = env.solve_pressure(env.u, env.v, np.zeros((21, 21)))
    assert pr.shape == (21, 21) and np.isfinite(pr).all()
    # export in the reference's on-disk layouts (target.npz keys u, v; NS_optmization.npz keys U, V, desired_*, actions)
    import tempfile, os
    from pde_control_gym import export
    with tempfile.TemporaryDirectory() as d:
        export.save_ns_target(os.path.join(d, "target.npz"), env)
        z = np.load(os.path.join(d, "target.npz"))
        assert sorted(z.files) == ["u", "v"] and z["u"].shape == (200, 21, 21) and z["u"].dtype == np.float64
        np.testing.assert_array_equal(z["u"][2], g["u2"])
        np.testing.assert_array_equal(export.load_ns_target(os.path.join(d, "target.npz")), env.U)
        export.save_ns_optimization(os.path.join(d, "res.npz"), env, Uref[..., 0], Uref[..., 1], list(g.actions[:2]))
        z = np.load(os.path.join(d, "res.npz"))
        assert sorted(z.files) == ["U", "V", "actions", "desired_U", "desired_V"] and z["V"].shape == (200, 21, 21)
        with pytest.raises(ValueError):
            export.save_ns_target(os.path.join(d, "x.npz"), np.zeros((3, 3)))


# ---- batched VecEnv ------------------------------------------------------------------------------------
def _vec(B, **over):
    import pde_control_gym
    p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, **over)          # nt=401, S=30 -> 14 steps/episode
    from pde_control_gym.src import TunedReward1D
    p["reward_class"] = TunedReward1D(400, -1e3, 3e2)
    rng = np.random.default_rng(0)
    p["reset_init_condition_func"] = lambda nx: np.ones(nx) * rng.uniform(1, 3)
    return pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, device="cpu", backend=FakeBackend(), **p)


def test_vecenv_sb3_semantics_autoreset_and_terminal_observation():
    B = 4
    env = _vec(B)
    assert env.num_envs == B and env.observation_space.shape == (100,) and env.action_space.shape == (1,)
    obs = env.reset()
    assert obs.shape == (B, 100) and obs.dtype == np.float32
    assert env.env_is_wrapped(object) == [False] * B and env.get_attr("nx") == [100] * B
    rng = np.random.default_rng(1)
    last = None
    for k in range(14):
        a = rng.uniform(-1, 1, (B, 1)).astype(np.float32)
        last = obs
        obs, rew, dones, infos = env.step(a)
        assert rew.shape == (B,) and dones.dtype == bool and len(infos) == B
        if k < 13:
            assert not dones.any() and all(i == {} for i in infos)
    assert dones.all()
    for i in range(B):
        assert infos[i]["terminal_observation"].shape == (100,) and infos[i]["TimeLimit.truncated"] is False
        # returned obs is the FIRST observation of the new episode: constant initial condition
        assert np.all(obs[i] == obs[i][0]) and 1 <= obs[i][0] <= 3
        assert not np.array_equal(infos[i]["terminal_observation"], obs[i])
    assert (env.core.time_index.numpy() == 0).all()
    obs, rew, dones, infos = env.step(rng.uniform(-1, 1, (B, 1)).astype(np.float32))
    assert not dones.any() and (env.core.time_index.numpy() == 30).all()


def test_gymnasium_vector_face_reshapes_the_sb3_face():
    import pde_control_gym
    a, b = _vec(3), pde_control_gym.GymnasiumVectorAdapter(_vec(3))
    o1 = a.reset()
    o2, info = b.reset(seed=1)
    assert info == {} and np.array_equal(o1, o2) and b.num_envs == 3 and b.single_action_space.shape == (1,)
    rng = np.random.default_rng(2)
    seen_done = False
    for k in range(16):
        act = rng.uniform(-1, 1, (3, 1)).astype(np.float32)
        o1, r1, d1, i1 = a.step(act)
        o2, r2, te, tr, i2 = b.step(act)
        assert np.array_equal(r1, r2) and np.array_equal(d1, te | tr) and not (te & tr).any()
        if d1.any():
            seen_done = True
            assert i2["_final_observation"].tolist() == d1.tolist()
            for i in np.nonzero(d1)[0]:
                np.testing.assert_array_equal(i2["final_observation"][i], i1[i]["terminal_observation"])
                assert te[i] != bool(i1[i]["TimeLimit.truncated"])
        else:
            assert i2 == {}
    assert seen_done


def test_vecenv_fused_autoreset_tensor_path():
    import torch
    B = 3
    env = _vec(B)
    env.reset_tensor()
    pool = np.stack([np.ones(100) * (7 + b) for b in range(B)]).astype(np.float32)
    env.enable_fused_auto_reset(pool)
    for k in range(14):
        obs, r, te, tr = env.step_tensor(torch.zeros(B))
    assert te.all()
    np.testing.assert_array_equal(obs.numpy(), pool)
    assert (env.core.time_index.numpy() == 0).all()
    assert not np.array_equal(env.core.t["final_obs"].numpy(), pool)


def test_vecenv_ns2d(golden_ns):
    import pde_control_gym
    from pde_control_gym.src import NSReward
    g = golden_ns["N1"]
    B = 2
    p = {"T": 0.2, "dt": 1e-3, "X": 1, "dx": 0.05, "Y": 1, "dy": 0.05, "action_dim": 1, "reward_class": NSReward(0.1),
         "normalize": False, "reset_init_condition_func": lambda X: (g.u0.copy(), g.v0.copy(), np.zeros_like(X)),
         "boundary_condition": NS_BC, "U_ref": np.zeros((200, 21, 21, 2)), "action_ref": 2.0 * np.ones(1000),
         "maximum_pressure_iteration": 50}
    env = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, device="cpu", backend=FakeBackend(),
                                   dtype="float64", **p)
    obs = env.reset()
    assert obs.shape == (B, 21, 21, 2)
    obs, rew, dones, infos = env.step(np.full((B, 1), 3.5))
    assert obs.shape == (B, 21, 21, 2) and rew.shape == (B,) and not dones.any()
    assert np.all(obs[:, -1, 1:-1, 0] == 3.5)


def test_reward_classes_host_definitions():
    """TunedReward1D.reward / NSReward.reward on caller-held arrays follow the reference formulas."""
    from pde_control_gym.src import NormReward, NSReward, TunedReward1D
    rng = np.random.default_rng(0)
    u = np.zeros((300, 10), dtype=np.float32)
    u[:151] = rng.uniform(-1, 1, (151, 10))
    rw = TunedReward1D(299, -1e3, 3e2)
    assert rw.reward(u, 150, False, False, 0.0) == np.linalg.norm(u[50]) - np.linalg.norm(u[150])
    assert rw.reward(u, 50, False, False, 0.0) == -np.linalg.norm(u[50])          # wraps into zero rows
    assert rw.reward(u, 150, False, True, 0.0) == -1e3 * (299 - 150)
    assert rw.reward(u, 150, True, False, 0.0) == 3e2 - np.sum(abs(u[:, -1])) / 1000 - np.linalg.norm(u[150])
    U = rng.uniform(-1, 1, (5, 4, 4, 2))
    Ur = rng.uniform(-1, 1, (5, 4, 4, 2))
    r = NSReward(0.1).reward(U, 2, Ur, 3.0, np.ones(5) * 2)
    assert r == pytest.approx(-0.5 * np.sum((U[2] - Ur[2]) ** 2) / 16 - 0.05)
    assert NormReward(10, "inf").reward(u, 3, False, False) == -np.abs(u[3]).max()
    with pytest.raises(Exception):
        NormReward()


def test_device_rollout_eager_matches_manual_loop():
    """DeviceRollout (eager path on the CPU test double) == stepping by hand with the same policy."""
    import torch
    from pde_control_gym import DeviceRollout
    B, T = 3, 6
    torch.manual_seed(0)
    pol = torch.nn.Sequential(torch.nn.Linear(100, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1), torch.nn.Tanh())
    env = _vec(B)
    env.reset_tensor()
    ro = DeviceRollout(env, pol, T, use_graph=False).run()
    env2 = _vec(B)
    obs = env2.reset_tensor()
    np.testing.assert_array_equal(ro.obs[0].numpy(), obs.numpy())
    for t in range(T):
        with torch.no_grad():
            a = pol(obs).reshape(B).clamp(-1, 1)
        obs, r, te, tr = env2.step_tensor(a)
        np.testing.assert_array_equal(ro.obs[t + 1].numpy(), obs.numpy())
        np.testing.assert_array_equal(ro.rewards[t].numpy(), r.numpy())
        np.testing.assert_array_equal(ro.actions[t].numpy(), a.numpy())


def test_device_rollout_with_fused_policy_and_exploration_noise_on_cpu_double():
    """Host logic of the fused-policy rollout (pdegym_mlp_forward stands in as the CPU double): the policy output plus the
    caller's noise, clamped, lands in the action buffer; the next run picks up in-place parameter updates (refresh)."""
    import torch
    from pde_control_gym import DeviceRollout
    from pdecontrolgym_amd.policy import FusedMLP
    from tests.fake_backend import FakeBackend
    B, T = 3, 5
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(100, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    env = _vec(B)
    env.reset_tensor()
    ro = DeviceRollout(env, FusedMLP(net, backend=FakeBackend()), T, use_graph=False, action_noise=True)
    ro.action_noise.normal_().mul_(0.3)
    ro.run()
    env2 = _vec(B)
    obs = env2.reset_tensor()
    for t in range(T):
        with torch.no_grad():
            a = (net(obs).reshape(B) + ro.action_noise[t]).clamp(-1, 1)
        np.testing.assert_allclose(ro.actions[t].numpy(), a.numpy(), rtol=2e-5, atol=2e-6)
        obs, r, te, tr = env2.step_tensor(ro.actions[t])          # follow the rollout's own actions: everything else is bitwise
        np.testing.assert_array_equal(ro.obs[t + 1].numpy(), obs.numpy())
        np.testing.assert_array_equal(ro.rewards[t].numpy(), r.numpy())
    before = ro.actions.clone()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(-1.0)
    ro.action_noise.zero_()
    ro.run()
    assert not torch.equal(before, ro.actions)
    with torch.no_grad():
        np.testing.assert_allclose(ro.actions[0].numpy(), net(ro.obs[0]).reshape(B).clamp(-1, 1).numpy(), rtol=2e-5, atol=2e-6)


def test_engine_rollout_equals_step_loop_on_the_cpu_double():
    """PDEBatch1D.rollout (pdegym_*_rollout on the GPU) is T step calls through the rollout buffers: host-side contract on the
    oracle-backed double, with fused auto-reset."""
    import torch
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    B, T, nx, S = 3, 7, 20, 5
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    kw = dict(T=3 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
              sensing_type=None, normalize=True, max_control_value=5.0, limit_pde_state_size=True, max_state_value=1e6)
    rng = np.random.default_rng(5)
    outs = []
    for mode in ("steps", "rollout"):
        e = PDEBatch1D("parabolic", reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B,
                       device="cpu", backend=FakeBackend(), **kw)
        n = e.n
        rng = np.random.default_rng(5)
        e.reset(torch.tensor(rng.uniform(1, 2, (B, n)).astype(np.float32)), torch.tensor(rng.uniform(-1, 1, (B, n)).astype(np.float32)))
        e.enable_auto_reset(torch.tensor(rng.uniform(1, 2, (2 * B, n)).astype(np.float32)), keep_final_obs=True)
        acts = torch.tensor(rng.uniform(-1, 1, (T, B)).astype(np.float32))
        obs = torch.zeros(T + 1, B, n)
        obs[0].copy_(e.t["obs"])
        rew, te, tr = torch.zeros(T, B), torch.zeros(T, B, dtype=torch.uint8), torch.zeros(T, B, dtype=torch.uint8)
        if mode == "steps":
            e.t["obs"] = obs[0]
            e.t["u"] = obs[0]
            for t in range(T):
                e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
        else:
            assert e.can_rollout()
            e.rollout(obs, acts, rew, te, tr)
            assert torch.equal(e.t["obs"], obs[T])
        outs.append((obs, rew, te, tr, e.t["time_index"].clone(), e.t["reset_count"].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[0][2].sum()) > 0


def test_bench_docstring_names_every_workload():
    """bench.py's module docstring is where the workloads are described: a workload added to WORKLOADS must be named there."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    missing = [k for k in bench.WORKLOADS if k not in (bench.__doc__ or "") and k != "parabolic_c2"]
    assert not missing, missing


@pytest.mark.parametrize("bk", BACKENDS)
@pytest.mark.parametrize("variant", ["separate_state", "collocated", "opposite"])
def test_device_rollout_consecutive_runs_start_from_the_last_observation(bk, variant):
    """Engines whose observation is NOT the state (state_in_obs=False, scalar sensing): after run() the engine's current
    observation must be slot T of the rollout, so that the next run() seeds obs[0] with it (advisor finding r3: it used to be the
    pre-rollout observation while ``u`` had advanced).  Two runs of T steps equal one run of 2T steps."""
    import torch
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    from pde_control_gym.src import TunedReward1D
    B, T = 4, 3
    extra = {"separate_state": dict(state_in_obs=False), "collocated": dict(sensing_loc="collocated", control_type="Neumann"),
             "opposite": dict(sensing_loc="opposite", sensing_type="Dirchilet")}[variant]

    def mk():
        rng = np.random.default_rng(11)
        p = _transport_params(T=0.2, dt=1e-4, control_sample_rate=20e-4, reward_class=TunedReward1D(2000, -1e3, 3e2))
        p.update({k: v for k, v in extra.items() if k != "state_in_obs"})
        p["reset_init_condition_func"] = lambda nx: np.linspace(1, 2, nx) * rng.uniform(1, 3)
        v = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_bk(bk), **{k: w for k, w in extra.items() if k == "state_in_obs"}, **p)
        v.reset_tensor()
        return v
    d = 100 if variant == "separate_state" else 1
    va, vb = mk(), mk()
    w = torch.linspace(-0.3, 0.4, d).to(va.device)                  # on the device up front: no host copy inside a graph capture
    pol = lambda o: torch.tanh(o.float() @ w)                       # noqa: E731
    assert not va.core.state_in_obs
    graph = bk == "hip"
    ro = DeviceRollout(va, pol, T, use_graph=graph)
    first = ro.run().obs.cpu().numpy().copy()
    np.testing.assert_array_equal(va.core.t["obs"].cpu().numpy(), first[T])
    second = ro.run().obs.cpu().numpy().copy()
    long = DeviceRollout(vb, pol, 2 * T, use_graph=graph).run().obs.cpu().numpy()
    np.testing.assert_array_equal(second[0], first[T])
    np.testing.assert_array_equal(np.concatenate([first, second[1:]]), long)


@pytest.mark.parametrize("bk", BACKENDS)
def test_truncation_reward_is_a_python_float_like_the_reference(golden_transport, bk):
    """tuned_reward_1d.py:38-39: the truncation branch is ``truncate_penalty * (nt - time_index)`` on the constructor's Python
    numbers -- the reference hands a Python float back (every other branch is NumPy arithmetic); so does the single environment
    (it used to be a np.float32)."""
    from pde_control_gym.src import TransportPDE1D, TunedReward1D
    g = golden_transport["R_trunc"]
    kw = dict(TRANSPORT_CASES["R_trunc"])
    ra = g.reward_args                                  # (nt, truncate_penalty, terminate_reward) as the examples pass them: Python numbers
    env = TransportPDE1D(reward_class=TunedReward1D(int(ra[0]), float(ra[1]), float(ra[2])), sensing_noise_func=lambda s: s,
                         reset_init_condition_func=lambda nx: g.init, reset_recirculation_func=lambda nx: g.beta, **_bk(bk), **kw)
    env.reset()
    for k in range(2):
        obs, r, te, tr, _ = env.step(np.float32(g.actions[k]))
        assert tr == bool(g.truncate[k])
        if tr:
            assert type(r) is float and r == float(g.reward[k]) == -8.0e6
        else:
            assert isinstance(r, np.floating)
            np.testing.assert_allclose(r, g.reward[k], rtol=1e-6)


@pytest.mark.parametrize("bk", BACKENDS)
def test_ns_step_past_the_last_frame_raises_index_error_like_the_reference(bk):
    """navier_stokes2D.py:147-148 writes U[time_index] after the increment, so the nt-th step of an episode (one past the terminal
    one) raises IndexError there; the drop-in used to keep going silently."""
    from pde_control_gym.src import NavierStokes2D, NSReward
    nt, n = 4, 11
    dx = 0.1
    dt = 0.2 * 0.5 * dx * dx / 0.1
    p = {"T": nt * dt, "dt": dt, "X": 1, "dx": dx, "Y": 1, "dy": dx, "action_dim": 1, "reward_class": NSReward(0.1), "normalize": False,
         "reset_init_condition_func": lambda X: (np.ones_like(X), np.zeros_like(X), np.zeros_like(X)), "boundary_condition": NS_BC,
         "U_ref": np.zeros((nt, n, n, 2)), "action_ref": 2.0 * np.ones(nt), "maximum_pressure_iteration": 5}
    env = NavierStokes2D(**_bk(bk), **p)
    assert env.nt == nt
    env.reset()
    flags = [env.step(3.0)[2] for _ in range(nt - 1)]
    assert flags == [False] * (nt - 2) + [True]
    with pytest.raises(IndexError, match=f"index {nt} is out of bounds for axis 0 with size {nt}"):
        env.step(3.0)
    env.reset()                                        # and the environment is usable again after a reset
    assert env.step(3.0)[2] is False


def test_roofline_counters_are_tied_to_the_kernel_sources(monkeypatch):
    """profiles/counters_latest.json carries, per workload, the fingerprint of the kernel sources its PMC counters were collected
    from; bench.py's roofline block says ``counters_stale`` when the tree has moved on.  The committed file must be
    stamped, every workload of the table must have its sources listed, and a changed source flips the flag."""
    import json
    import bench

    class W:
        def algorithmic_bytes_per_step(self):
            return 1

        def compulsory_bytes_per_step(self):
            return 1
    d = json.load(open(os.path.join(ROOT, "profiles", "counters_latest.json")))
    assert set(d["workloads"]) <= set(bench.KERNEL_SOURCES) and set(bench.WORKLOADS) == set(bench.KERNEL_SOURCES)
    assert all(v.get("kernel_stamp") for v in d["workloads"].values())
    r = bench.roofline_block(W(), "parabolic_c2", 0.02, True)
    assert r["counters_stale"] == (r["counters_kernel_stamp"] != bench.kernel_stamp("parabolic_c2")) and 0.3 < r["frac"] < 0.7
    monkeypatch.setattr(bench, "kernel_stamp", lambda k: "0" * 16)
    assert bench.roofline_block(W(), "parabolic_c2", 0.02, True)["counters_stale"] is True
    # the fingerprint follows the CODE: comments and white space apart (the public header is part of every workload's fingerprint,
    # and a reworded comment there must not make every committed counter look stale); string literals are code
    from pdecontrolgym_amd import build
    base = b'int a = 1;  /* x */ const char* s = "// kept /* kept */";\n// tail\n'
    assert build._code_only(base) == build._code_only(b'int a = 1; const char* s = "// kept /* kept */"; // other words\n/* more */')
    assert build._code_only(base) != build._code_only(base.replace(b"a = 1", b"a = 2"))
    assert build._code_only(base) != build._code_only(base.replace(b"// kept", b"// changed"))


@pytest.mark.parametrize("control,loc", [("Neumann", "full"), ("Dirchilet", "collocated"), ("Neumann", "collocated")])
def test_engine_rollout_contract_for_neumann_and_scalar_sensing_on_the_double(control, loc):
    """Host logic of PDEBatch1D.rollout for the general cases (round 4): with full-state sensing the observation slots carry the
    state, with scalar sensing the engine's own ``u`` does and the slots are [T + 1, B, 1]; either way T step calls and one
    rollout call leave identical outputs and identical engine state (the CPU double runs the oracle per step)."""
    import torch
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    B, T, nx, S = 3, 7, 40, 5
    dx = 1.0 / nx
    dt = 0.5 * dx
    kw = dict(T=3 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type=control, sensing_loc=loc, sensing_type=None,
              normalize=True, max_control_value=2.0, limit_pde_state_size=True, max_state_value=1e6)
    rng = np.random.default_rng(3)
    init, beta = rng.uniform(0.5, 2, (B, nx)).astype(np.float32), rng.uniform(-1, 1, (B, nx)).astype(np.float32)
    pool = rng.uniform(0.5, 2, (2 * B, nx)).astype(np.float32)
    acts = torch.tensor(rng.uniform(-1, 1, (T, B)).astype(np.float32))
    outs = []
    for mode in ("steps", "rollout"):
        e = PDEBatch1D("transport", reward=RewardSpec(N.REWARD_TUNED1D, int(round(kw["T"] / dt)), -1e3, 3e2), num_envs=B, device="cpu",
                       backend=FakeBackend(), **kw)
        assert e.can_rollout()
        e.reset(torch.tensor(init), torch.tensor(beta))
        e.enable_auto_reset(torch.tensor(pool))
        od = e.obs_dim
        obs = torch.zeros(T + 1, B, od)
        obs[0].copy_(e.t["obs"])
        rew, te, tr = torch.zeros(T, B), torch.zeros(T, B, dtype=torch.uint8), torch.zeros(T, B, dtype=torch.uint8)
        if mode == "steps":
            if e.state_in_obs:
                e.t["obs"] = e.t["u"] = obs[0]
            for t in range(T):
                e.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
        else:
            e.rollout(obs, acts, rew, te, tr)
            with pytest.raises(ValueError):
                e.rollout(obs, acts, rew, te, tr, obs_noise=torch.zeros(T, B, od))        # noise shapes a POLICY's input
        outs.append([x.numpy().copy() for x in (obs, rew, te, tr, e.t["u"], e.t["time_index"], e.t["bsum"], e.t["reset_count"])])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert outs[0][2].sum() > 0


def test_ns_engine_rollout_contract_on_the_double():
    """NSBatch2D.rollout (pdegym_ns2d_rollout_*, round 4): T step calls and one rollout call leave identical outputs and engine
    state, fused auto-reset included; grids outside the column kernel's set refuse (host rule = the C ABI's)."""
    import torch
    from pdecontrolgym_amd.batch2d import NSBatch2D
    n, nt, K, B, T = 11, 4, 3, 2, 6
    rng = np.random.default_rng(8)
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=NS_BC, U_ref=rng.uniform(-1, 1, (nt, n, n, 2)),
              action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K)
    first = [rng.uniform(-1, 1, (B, n, n)) for _ in range(3)]
    pools = [rng.uniform(-1, 1, (2 * B, n, n)) for _ in range(3)]
    acts = torch.as_tensor(rng.uniform(2, 4, (T, B, 1)))
    outs = []
    for mode in ("steps", "rollout"):
        env = NSBatch2D(num_envs=B, device="cpu", dtype=torch.float64, backend=FakeBackend(), **kw)
        assert env.can_rollout()
        env.reset(*first)
        env.enable_auto_reset(*pools)
        obs = torch.zeros(T + 1, B, n, n, 2, dtype=torch.float64)
        obs[0].copy_(env.t["obs"])
        rew, te = torch.zeros(T, B, dtype=torch.float64), torch.zeros(T, B, dtype=torch.uint8)
        if mode == "steps":
            env.t["obs"] = obs[0]
            for t in range(T):
                env.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t])
        else:
            env.rollout(obs, acts, rew, te)
        outs.append([x.numpy().copy() for x in (obs, rew, te, env.p, env.t["time_index"], env.t["reset_count"], env.t["final_obs"])])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert outs[0][2].sum() >= 2
    odd = dict(kw, X=1.2, Y=1.2, U_ref=np.zeros((nt, 13, 13, 2)))                  # 13 rows: not one of the column kernel's heights
    assert not NSBatch2D(num_envs=1, device="cpu", dtype=torch.float64, backend=FakeBackend(), **odd).can_rollout()


@pytest.mark.parametrize("hook", ["numpy", "tensor"])
@pytest.mark.parametrize("fused", [False, True])
def test_sensing_noise_hooks_also_see_auto_reset_and_terminal_observations(hook, fused):
    """The reference applies sensing_noise_func to what step() returns (hyperbolic.py:160-164) AND to what reset() returns (:224), so
    on the batched face the hook must also cover the first observation of an auto-reset instance and the terminal observation
    (advisor finding r3: with f = obs + 100 a reset gave mean 101, a normal step 100.8, the step that auto-resets 1.0)."""
    import torch
    B = 3
    kw = {"sensing_noise_func": (lambda o: o + 100.0)} if hook == "numpy" else {"sensing_noise_func": None,
                                                                                 "sensing_noise_tensor_func": (lambda o: o + 100.0)}
    env = _vec(B, **kw)                                # episodes of 14 env-steps
    obs = env.reset()
    assert obs.min() >= 100
    if fused:
        env.enable_fused_auto_reset()
    a = np.zeros((B, 1), np.float32)
    for k in range(14):
        obs, rew, dones, infos = env.step(a)
        assert obs.min() >= 100, (k, obs.min())        # incl. the step that restarts every instance (un-noised values are < 10)
    assert dones.all()
    for i in range(B):
        assert infos[i]["terminal_observation"].min() >= 100 and infos[i]["terminal_observation"].max() < 100 + 1e4


def test_integration_md_binding_stub_matches_the_abi():
    """The ctypes structures INTEGRATION.md section 2 shows a reference maintainer are executed as written and compared with the
    bindings this repo uses (field names, order, size): a stub that lags an ABI bump would corrupt memory for whoever copies it."""
    import ctypes as C
    from pdecontrolgym_amd import _native as N
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ns = {}
    for name, end, real in (("Params1D", "class Bufs1D", N.Params1D), ("Bufs1D", "def reset(self", N.Bufs1D),
                            ("Rollout1D", "ro = Rollout1D", N.Rollout1D)):
        i = text.index(f"class {name}(C.Structure):")
        exec("import ctypes as C\n" + text[i:text.index(end, i)], ns)
        stub = ns[name]
        assert [f[0] for f in stub._fields_] == [f[0] for f in real._fields_], name
        assert C.sizeof(stub) == C.sizeof(real), name


@pytest.mark.parametrize("kind", ["double", pytest.param("hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("which", range(5), ids=["transport", "parabolic", "navier_stokes", "traffic", "brain_tumor"])
def test_every_registered_id_builds_resets_and_steps_from_its_reference_parameter_dictionary(kind, which):
    """The five parameter dictionaries tests/test_real_sb3.py hands to the REAL ``gymnasium.make`` + ``check_env`` (skipped in this
    image), through ``pde_control_gym.make``: spaces, reset and step types of the Env contract, checked by hand."""
    import pde_control_gym
    from tests.five_ids import _backend, _five_ids
    env_id, params = _five_ids()[which]
    env = pde_control_gym.make(env_id, **_backend(kind), **params)
    e = env.unwrapped
    obs, info = env.reset(seed=0)
    assert isinstance(info, dict) and np.asarray(obs).shape == e.observation_space.shape
    assert np.asarray(obs).dtype == e.observation_space.dtype or which in (2, 4)        # NS / tumour observe in float64 like the reference
    a = e.action_space.sample()
    out = env.step(a if which != 4 else float(np.asarray(a).reshape(-1)[0]))
    assert len(out) == 5
    o, r, te, tr, inf = out
    assert np.asarray(o).shape == e.observation_space.shape and np.isfinite(float(r)) and isinstance(inf, dict)
    assert isinstance(te, (bool, np.bool_)) and isinstance(tr, (bool, np.bool_))


def test_host_io_mode_of_the_1d_engine_is_the_batch_of_one_face_and_equals_the_staged_path():
    """PDEBatch1D.enable_host_io (the single environments' hand-over through one pinned allocation): refused for engines whose
    observation IS the state and for batches; with it, step_host() returns what step() returns (CPU double: the same tensors, plain
    host memory) for float32 and float64 commands."""
    import torch
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    kw = dict(T=0.3, dt=1e-3, X=1, dx=0.02, control_sample_rate=5e-3, normalize=True, limit_pde_state_size=True, max_control_value=20)
    mk = lambda **extra: PDEBatch1D("transport", reward=RewardSpec(N.REWARD_TUNED1D, 300, -1e3, 3e2), device="cpu", backend=FakeBackend(),
                                    **dict(kw, **extra))
    with pytest.raises(ValueError, match="state in its own tensor"):
        mk(num_envs=1).enable_host_io()
    with pytest.raises(ValueError, match="batch-of-one"):
        mk(num_envs=2, state_in_obs=False).enable_host_io()
    a, b = mk(num_envs=1, state_in_obs=False), mk(num_envs=1, state_in_obs=False)
    io = a.enable_host_io()
    assert a.enable_host_io() is io                                  # idempotent
    init, beta = np.linspace(1, 2, 50, dtype=np.float32)[None], np.ones(50, dtype=np.float32)
    a.reset(init, beta)
    b.reset(init, beta)
    for k, (val, kind) in enumerate([(0.25, N.ACTION_F32), (-0.5, N.ACTION_F64), (0.125, N.ACTION_WEAK), (0.75, N.ACTION_F32)]):
        a.step_host(val, kind)
        o, r, te, tr = b.step(torch.tensor([val], dtype=torch.float32 if kind == N.ACTION_F32 else torch.float64), action_kind=kind)
        np.testing.assert_array_equal(io["obs"], o.numpy())
        np.testing.assert_array_equal(io["reward"], r.numpy())
        assert int(io["terminated"][0]) == int(te[0]) and int(io["truncated"][0]) == int(tr[0])
        np.testing.assert_array_equal(a.u.numpy(), b.u.numpy())
        assert int(a.time_index[0]) == int(b.time_index[0]) == 5 * (k + 1)


def test_prepared_step_calls_switch_to_the_state_device(monkeypatch):
    """HipBackend.prepare_step1d / prepare_ns2d_step (the batch-of-one faces' launches with pre-built argument structures) carry the
    device guard INSIDE the prepared call: entered with the device of the state tensors when another device is current, not at all
    when it is current already -- unit-tested with stand-ins like the decorator form above (no GPU, no library)."""
    import ctypes as C
    import torch
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd import backend as bk_mod
    entered, launched = [], []

    class Guard:
        def __init__(self, dev):
            entered.append(dev)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class Stream:
        cuda_stream = 1234

    current = {"dev": 0}
    monkeypatch.setattr(torch.cuda, "device", Guard)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: current["dev"])
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: Stream())

    class T:                       # a tensor stand-in on cuda:1
        device = torch.device("cuda", 1)
        dtype = torch.float32

        def dim(self):
            return 2

    class Lib:
        @staticmethod
        def pdegym_transport_step(p, b, B, stream):
            launched.append((B, stream))
            return 0
        pdegym_parabolic_step = pdegym_transport_step
        pdegym_ns2d_step_f32 = pdegym_transport_step

    bk = bk_mod.HipBackend.__new__(bk_mod.HipBackend)
    bk.lib = Lib()
    monkeypatch.setattr(bk_mod.HipBackend, "_bufs1d", staticmethod(lambda Td: N.Bufs1D()))
    monkeypatch.setattr(bk_mod.HipBackend, "_bufs_ns", staticmethod(lambda Td, dt: N.BufsNS2D()))
    P = N.Params1D()
    call = bk.prepare_step1d("transport", P, {"beta": T(), "bsum": T()}, 1)
    call()
    assert entered == [torch.device("cuda", 1)] and launched == [(1, 1234)]
    current["dev"] = 1                      # the state's device is current: no context manager
    call()
    assert entered == [torch.device("cuda", 1)] and len(launched) == 2
    current["dev"] = 0
    ns = bk.prepare_ns2d_step(N.ParamsNS2D(), {"p": T()}, 1)
    ns()
    assert entered[-1] == torch.device("cuda", 1) and len(entered) == 2 and len(launched) == 3
