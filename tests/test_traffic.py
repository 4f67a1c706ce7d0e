"""TrafficPDE1D: host API on the CPU test double, and GPU parity through the C ABI.

GPU bars: r, y, v and observations BIT-EXACT float64 vs the reference's golden vectors and vs the oracle; rewards rtol
1e-13 (wave butterfly vs BLAS ddot order); flags equal."""
import contextlib
import io
import random

import numpy as np
import pytest

from tests.fake_backend import FakeBackend
from tests.test_oracle_golden import TRAFFIC_CASES

torch = pytest.importorskip("torch")
BASE = dict(T=240, dt=0.25, X=500, dx=10, v_steady=10, ro_steady=0.12, v_max=40, ro_max=0.16, tau=60)


def _single(g, **extra):
    from pde_control_gym.src import TrafficARZReward, TrafficPDE1D
    random.seed(11)
    env = TrafficPDE1D(reward_class=TrafficARZReward(), simulation_type=str(g.sim), limit_pde_state_size=bool(g.limit),
                       control_freq=int(g.control_freq), **BASE, **extra)
    random.seed(13)
    return env


def _check_single(env, g):
    obs, info = env.reset()
    assert info == {} and obs.shape == (102,) and obs.dtype == np.float64
    np.testing.assert_array_equal(obs, g.obs[0])
    assert env.rs == float(g.rs)
    keep = {int(k): i for i, k in enumerate(g.keep)} if "keep" in g else None
    for k, a in enumerate(g.actions):
        obs, r, d, t, info = env.step(a)
        assert isinstance(d, bool) and isinstance(t, bool)
        if keep is None:
            np.testing.assert_array_equal(obs, g.obs[k + 1], err_msg=f"step {k}")
        elif (k + 1) in keep:
            np.testing.assert_array_equal(obs, g.obs[keep[k + 1]], err_msg=f"step {k}")
        np.testing.assert_allclose(r, g.reward[k], rtol=1e-13)
        assert d == bool(g.done[k]) and t == bool(g.trunc[k]) and env.time_index == g.time[k]
    assert env.r.shape == (51, 1) and env.v.shape == (51, 1)


@pytest.mark.parametrize("case", TRAFFIC_CASES)
def test_traffic_public_api_on_test_double(golden_traffic, case):
    _check_single(_single(golden_traffic[case], device="cpu", backend=FakeBackend()), golden_traffic[case])


def test_traffic_constructor_errors():
    from pde_control_gym.src import TrafficARZReward, TrafficPDE1D
    kw = dict(reward_class=TrafficARZReward(), device="cpu", backend=FakeBackend(), **BASE)
    with pytest.raises(ValueError, match="Invalid simulation type"):
        TrafficPDE1D(simulation_type="sideways", **kw)
    with pytest.raises(ValueError, match="equilibrium condition"):
        TrafficPDE1D(simulation_type="inlet", **dict(kw, v_steady=11))
    with pytest.raises(AssertionError, match="control_freq must be a positive integer"):
        TrafficPDE1D(simulation_type="inlet", control_freq=0, **kw)
    env = TrafficPDE1D(simulation_type="both", **kw)
    assert env.action_space.shape == (2,) and env.observation_space.shape == (102,) and env.M == 51
    assert env.action_space.low[0] == env.qs * 0.8


@pytest.mark.gpu
@pytest.mark.parametrize("case", TRAFFIC_CASES)
def test_traffic_hip_matches_reference_golden(golden_traffic, case):
    _check_single(_single(golden_traffic[case]), golden_traffic[case])


@pytest.mark.gpu
@pytest.mark.parametrize("sim,cf,B", [("inlet", 1, 37), ("outlet", 2, 64), ("both", 1, 5), ("outlet-train", 3, 130)])
def test_traffic_hip_matches_oracle_batched(sim, cf, B):
    """Per-instance steady states / actions, masked reset mid-run: every field bit-exact vs the oracle."""
    from oracle import pde_oracle as po
    from pdecontrolgym_amd.batch_traffic import TrafficBatch
    rng = np.random.default_rng(B)
    rs = rng.choice([0.115, 0.12, 0.125], B)
    orc = po.TrafficOracle(240, 0.25, 500, 10, sim, 40, 0.16, 60, True, cf)
    env = TrafficBatch(240, 0.25, 500, 10, sim, 40, 0.16, 60, True, cf, num_envs=B, device="cuda")
    qclip = rng.choice([0.115, 0.12, 0.125], B)
    qclip = qclip * (40 * (1 - qclip / 0.16))
    env.set_action_bounds(qclip)
    o_ref = orc.reset(rs, qclip)
    o = env.reset(rs)
    np.testing.assert_array_equal(o.cpu().numpy(), o_ref)
    nact = 2 if sim == "both" else 1
    for k in range(25):
        a = rng.uniform(0.7, 1.3, (B, nact)) * orc.qs[:, None]
        o_ref, r_ref, d_ref, t_ref = orc.step(a)
        o, r, d, t = env.step(a)
        np.testing.assert_array_equal(o.cpu().numpy(), o_ref, err_msg=f"step {k}")
        np.testing.assert_array_equal(env.t["r"].cpu().numpy(), orc.r)
        np.testing.assert_array_equal(env.t["y"].cpu().numpy(), orc.y)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-13)
        np.testing.assert_array_equal(d.cpu().numpy().astype(bool), d_ref)
        np.testing.assert_array_equal(t.cpu().numpy().astype(bool), t_ref)
        np.testing.assert_array_equal(env.t["time"].cpu().numpy(), orc.time_index)
    # masked reset: only the selected instances restart (with a new steady state)
    mask = (np.arange(B) % 3 == 0)
    rs2 = np.where(mask, 0.125, rs)
    env.reset(rs2, mask=torch.tensor(mask.astype(np.uint8)))
    fresh = po.TrafficOracle(240, 0.25, 500, 10, sim, 40, 0.16, 60, True, cf)
    fresh.reset(rs2, qclip)
    np.testing.assert_array_equal(env.t["r"].cpu().numpy()[mask], fresh.r[mask])
    np.testing.assert_array_equal(env.t["r"].cpu().numpy()[~mask], orc.r[~mask])
    assert (env.t["time"].cpu().numpy()[mask] == 0).all() and (env.t["time"].cpu().numpy()[~mask] > 0).all()


def test_traffic_vecenv_on_test_double():
    import pde_control_gym
    from pde_control_gym.src import TrafficARZReward
    random.seed(0)
    venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=5, device="cpu", backend=FakeBackend(),
                                    reward_class=TrafficARZReward(), simulation_type="outlet-train", limit_pde_state_size=True,
                                    control_freq=2, **BASE)
    obs = venv.reset()
    assert obs.shape == (5, 102) and obs.dtype == np.float64
    qs = venv.core.t["qs_clip"].numpy()
    obs, rew, dones, infos = venv.step(qs[:, None] * 1.05)
    assert obs.shape == (5, 102) and rew.shape == (5,) and not dones.any()
    assert venv.observation_space.shape == (102,) and venv.action_space.shape == (1,)


@pytest.mark.parametrize("hip", [pytest.param(False, id="cpu-double"), pytest.param(True, id="hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("sim", ["outlet", "outlet-train"])
def test_traffic_custom_reward_on_both_faces(sim, hip):
    """A user BaseReward subclass on TrafficPDE1D (docs/source/utils/customrewards.rst; called as traffic_arz_env.py:228 calls
    it): restating TrafficARZReward must reproduce the in-kernel reward AND the done rule that hangs on it (:233), on the single
    environment and on the batched face; a different reward changes the rewards only."""
    import pde_control_gym
    from pde_control_gym.src import BaseReward, TrafficARZReward, TrafficPDE1D
    bk = dict(device="cuda") if hip else dict(device="cpu", backend=FakeBackend())

    class Again(BaseReward):
        def reward(self, v_desired, r_desired, v, r):
            assert v.shape == (51, 1) and r.shape == (51, 1)
            return -(np.linalg.norm(v - v_desired) / v_desired + np.linalg.norm(r - r_desired) / r_desired)

    class Other(BaseReward):
        def reward(self, v_desired, r_desired, v, r):
            return -float(np.abs(r - r_desired).max()) - 1.0

    kw = dict(simulation_type=sim, limit_pde_state_size=True, control_freq=2, **BASE)
    B = 4
    envs = {}
    for name, rc in (("kernel", TrafficARZReward()), ("again", Again()), ("other", Other())):
        random.seed(3)
        envs[name] = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=B, reward_class=rc, **bk, **kw)
        random.seed(4)
        envs[name].reset()
    random.seed(3)
    single = TrafficPDE1D(reward_class=Again(), **bk, **kw)
    random.seed(3)
    single_k = TrafficPDE1D(reward_class=TrafficARZReward(), **bk, **kw)
    single.reset()
    single_k.reset()
    qs = envs["kernel"].core.t["qs_clip"].cpu().numpy()
    for i in range(4):
        a = qs[:, None] * (0.9 + 0.05 * i)
        ok, rk, dk, _ = envs["kernel"].step(a)
        oa, ra, da, _ = envs["again"].step(a)
        oo, ro, do, _ = envs["other"].step(a)
        np.testing.assert_array_equal(ok, oa)
        np.testing.assert_array_equal(ok, oo)
        np.testing.assert_allclose(ra, rk, rtol=1e-6)
        np.testing.assert_array_equal(da, dk)
        assert (ro <= -1.0).all()
        if sim != "outlet-train":
            assert not do.any()                # the other reward never exceeds -0.00023, and time is far from T
        s1 = single.step(np.array([single.qs * (0.9 + 0.05 * i)]))
        s2 = single_k.step(np.array([single_k.qs * (0.9 + 0.05 * i)]))
        np.testing.assert_array_equal(s1[0], s2[0])
        np.testing.assert_allclose(s1[1], s2[1], rtol=1e-12)
        assert s1[2] == s2[2] and s1[3] == s2[3]


@pytest.mark.gpu
def test_traffic_device_rollout_with_fused_policy():
    """DeviceRollout on TrafficPDE1D with FusedMLP reading the float64 observation (102 entries) and writing the float64
    outlet command (policy output scaled into the action box by the clamp): same rollout as the wrapped torch module within
    the float32 agreement of the forward passes; the graph replay equals the eager run bit for bit."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout, FusedMLP
    from pde_control_gym.src import TrafficARZReward
    torch = pytest.importorskip("torch")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(102, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1)).cuda()
    with torch.no_grad():
        net[-1].bias.fill_(4.6)                     # around the steady-state flow q_s
    runs = {}
    for name, pol, graph in (("torch", lambda o: net(o.float()).double(), False), ("fused", FusedMLP(net), False),
                             ("fused_graph", FusedMLP(net), True)):
        random.seed(0)
        venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=96, reward_class=TrafficARZReward(),
                                        simulation_type="outlet", limit_pde_state_size=True, control_freq=2, **BASE)
        venv.reset_tensor()
        ro = DeviceRollout(venv, pol, 8, use_graph=graph, action_low=3.0, action_high=6.0).run()
        torch.cuda.synchronize()
        runs[name] = [x.cpu().numpy().copy() for x in (ro.actions, ro.obs, ro.rewards)]
    assert runs["fused"][0].dtype == np.float64 and runs["fused"][0].std() > 0
    for got, want in zip(runs["fused"], runs["torch"]):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-7)
    for got, want in zip(runs["fused_graph"], runs["fused"]):
        np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("sim,cf,B,T", [("inlet", 1, 37, 12), ("outlet", 2, 64, 9), ("both", 1, 5, 30), ("outlet-train", 3, 130, 7),
                                        ("outlet", 2, 1, 1),
                                        # round 5: launches longer than one command batch (64 env-steps per load; one and two commands)
                                        ("both", 1, 3, 130), ("outlet", 2, 4, 65), ("inlet", 1, 2, 64), ("outlet-train", 2, 5, 129)])
def test_traffic_rollout_kernel_equals_step_calls_bitwise(sim, cf, B, T):
    """pdegym_traffic_rollout (T env-steps in one launch, (r, y) in registers across steps) against T step calls: every
    observation slot, reward, flag and the final r, y, time agree bit for bit."""
    from pdecontrolgym_amd.batch_traffic import TrafficBatch
    rng = np.random.default_rng(B + T)
    rs = rng.choice([0.115, 0.12, 0.125], B)
    qclip = rs * (40 * (1 - rs / 0.16))
    nact = 2 if sim == "both" else 1
    acts = torch.tensor(rng.uniform(0.7, 1.3, (T, B, nact)) * qclip[None, :, None], device="cuda")
    res = []
    for mode in ("steps", "rollout"):
        env = TrafficBatch(0.5, 0.25, 500, 10, sim, 40, 0.16, 60, True, cf, num_envs=B, device="cuda")   # T/dt = 2 s: episodes end inside
        env.set_action_bounds(qclip)
        env.reset(rs)
        if B % 2 == 1:                                     # odd batches: with the fused auto-reset (pool of 2B + 1 steady states)
            env.enable_auto_reset(np.random.default_rng(3).choice([0.115, 0.12, 0.125], 2 * B + 1))
        obs = torch.zeros(T + 1, B, 2 * env.M, dtype=torch.float64, device="cuda")
        obs[0].copy_(env.t["obs"])
        rew = torch.zeros(T, B, dtype=torch.float64, device="cuda")
        dn = torch.zeros(T, B, dtype=torch.uint8, device="cuda")
        tr = torch.zeros(T, B, dtype=torch.uint8, device="cuda")
        if mode == "steps":
            for t in range(T):
                o, r, d, tc = env.step(acts[t])
                obs[t + 1].copy_(o), rew[t].copy_(r), dn[t].copy_(d), tr[t].copy_(tc)
        else:
            assert env.can_rollout()
            env.rollout(obs, acts, rew, dn, tr)
        res.append([x.cpu().numpy().copy() for x in (obs, rew, dn, tr, env.t["r"], env.t["y"], env.t["time"], env.t["obs"], env.t["reward"],
                                                     env.t["rs"])]
                   + [env.t[k].cpu().numpy().copy() for k in ("reset_count", "final_obs") if env.t.get(k) is not None])
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    assert T < 9 or res[0][2].sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("sim,sizes", [("outlet", [102, 32, 1]), ("both", [102, 64, 64, 2]), ("outlet-train", [102, 1]),
                                       ("both", [102, 256, 256, 2]), ("inlet", [102, 100, 1]), ("outlet-train", [102, 48, 130, 1])])
def test_traffic_one_launch_rollout_with_policy_inside(sim, sizes):
    """DeviceRollout on TrafficPDE1D as ONE kernel (policy evaluated inside pdegym_traffic_rollout on the float64 observation
    rounded to float32) against policy launch + step launch per env-step: commands to float32 rounding, trajectories follow.
    A layer of more than 64 units takes the cooperative MFMA evaluation: everything equal bit for bit."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout, FusedMLP
    from pde_control_gym.src import TrafficARZReward
    torch.manual_seed(1)
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(torch.nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            mods.append(torch.nn.Tanh())
    net = torch.nn.Sequential(*mods).cuda()
    with torch.no_grad():
        net[-1].bias.fill_(4.6 if sim != "outlet-train" else 4.0)
    T, B = 10, 70
    runs = {}
    for mode in (False, True):
        random.seed(0)
        venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=B, reward_class=TrafficARZReward(),
                                        simulation_type=sim, limit_pde_state_size=True, control_freq=2, **BASE)
        venv.reset_tensor()
        ro = DeviceRollout(venv, FusedMLP(net), T, action_low=3.0, action_high=6.0, action_noise=True, one_launch=mode)
        assert ro.one_launch == mode
        ro.action_noise.copy_(torch.randn(ro.action_noise.shape, generator=torch.Generator().manual_seed(2)).mul(0.05).cuda())
        ro.run()
        torch.cuda.synchronize()
        runs[mode] = [x.cpu().numpy().copy() for x in (ro.actions, ro.obs, ro.rewards, ro.terminated, ro.truncated, venv.core.t["obs"],
                                                       venv.core.t["time"])]
    a, b = runs[True], runs[False]
    assert a[0].dtype == np.float64 and a[0].std() > 0
    if max(sizes[1:]) > 64:
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_allclose(a[0][0], b[0][0], rtol=2e-6, atol=1e-6)       # same observation: forward passes only
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-6)
    for x, y in zip(a[3:5], b[3:5]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(a[6], b[6])
    np.testing.assert_array_equal(a[5], a[1][T])


@pytest.mark.gpu
@pytest.mark.parametrize("sim,X,cf,B", [("outlet-train", 500, 2, 37), ("inlet", 500, 1, 64), ("both", 1500, 1, 5), ("outlet", 630, 3, 9)])
def test_traffic_fused_auto_reset_equals_host_driven_reset(sim, X, cf, B):
    """bufs.reset_rs: an instance whose step ends done | truncated restarts inside the launch (TrafficPDE1D.reset: new steady
    state from the pool, sine profile, time 0) -- bit for bit what a masked pdegym_traffic_reset_masked after the step does;
    final_obs keeps the terminal observation; the register kernel (M = 51, 64) and the LDS one (M = 151)."""
    from pdecontrolgym_amd.batch_traffic import TrafficBatch
    rng = np.random.default_rng(B)
    rs = rng.choice([0.115, 0.12, 0.125], B)
    pool = rng.choice([0.115, 0.12, 0.125], 2 * B + 3)
    qclip = rs * (40 * (1 - rs / 0.16))
    nact = 2 if sim == "both" else 1
    envs = [TrafficBatch(0.5, 0.25, X, 10, sim, 40, 0.16, 60, True, cf, num_envs=B, device="cuda") for _ in range(2)]   # T/dt = 2 s
    for e in envs:
        e.set_action_bounds(qclip)
        e.reset(rs)
    envs[0].enable_auto_reset(pool)
    cnt = np.zeros(B, dtype=np.int64)
    cur_rs = rs.copy()
    ends = 0
    for k in range(30):
        a = torch.tensor(rng.uniform(0.7, 1.3, (B, nact)) * qclip[:, None], device="cuda")
        o0, r0, d0, t0 = envs[0].step(a)
        o1, r1, d1, t1 = envs[1].step(a)
        fin = (d1.cpu().numpy() | t1.cpu().numpy()).astype(bool)
        np.testing.assert_array_equal(r0.cpu().numpy(), r1.cpu().numpy())
        np.testing.assert_array_equal(d0.cpu().numpy(), d1.cpu().numpy())
        np.testing.assert_array_equal(t0.cpu().numpy(), t1.cpu().numpy())
        last = o1.cpu().numpy().copy()
        if fin.any():
            ends += int(fin.sum())
            np.testing.assert_array_equal(envs[0].t["final_obs"].cpu().numpy()[fin], last[fin])
            cur_rs = np.where(fin, pool[(np.arange(B) + cnt * B) % len(pool)], cur_rs)
            cnt += fin
            o1 = envs[1].reset(cur_rs, mask=torch.tensor(fin.astype(np.uint8)))
        for key in ("r", "y", "time", "rs"):
            np.testing.assert_array_equal(envs[0].t[key].cpu().numpy(), envs[1].t[key].cpu().numpy(), err_msg=f"{key} step {k}")
        np.testing.assert_array_equal(o0.cpu().numpy(), o1.cpu().numpy(), err_msg=f"obs step {k}")
        np.testing.assert_array_equal(envs[0].t["reset_count"].cpu().numpy(), cnt)
    assert ends > 0


def test_traffic_vecenv_fused_auto_reset_on_test_double():
    import pde_control_gym
    from pde_control_gym.src import TrafficARZReward
    random.seed(0)
    kw = dict(BASE, T=0.5)
    venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=4, device="cpu", backend=FakeBackend(),
                                    reward_class=TrafficARZReward(), simulation_type="outlet-train", limit_pde_state_size=True,
                                    control_freq=1, **kw)
    venv.reset()
    venv.enable_fused_auto_reset()
    qs = venv.core.t["qs_clip"].numpy()
    seen = 0
    for _ in range(12):
        obs, rew, dones, infos = venv.step(qs[:, None] * 1.05)
        for i in np.nonzero(dones)[0]:
            seen += 1
            assert "terminal_observation" in infos[i] and infos[i]["terminal_observation"].shape == (102,)
            assert venv.core.t["time"][i].item() == 0.0
    assert seen > 0 and int(venv.core.t["reset_count"].sum()) == seen
