"""GPU tests of the drop-in Python API (pde_control_gym.*) on the real HIP backend, including the
reference's PUBLISHED known answers reproduced in closed loop (backstepping controller episodes; notebook
stored outputs cited in SURVEY.md section 6 / BASELINE.md)."""
import os
import sys

import numpy as np
import pytest

from tests.cases import NS_BC
from tests.test_host_api import _transport_params
from tests.test_oracle_golden import KAT_PUBLISHED

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loaded_backend_is_the_hip_library():
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.backend import default_backend
    b = default_backend()
    assert b.name == "hip-gfx950" and b.lib.pdegym_abi_version() == N.ABI_VERSION


@pytest.mark.parametrize("name", sorted(KAT_PUBLISHED))
def test_published_known_answers_closed_loop_on_gpu(golden_kat, name):
    """transport1Dbackstepping.py / reactionDiffusion1DBackstepping.py episodes through gym.make-style envs:
    episode reward and sum of L2 norms match the notebook values to rtol 1e-5 (they were produced with NumPy 1.26
    float64 scalar accumulation; fields themselves are bit-exact per step)."""
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    g = golden_kat[name]
    u0 = 1.0 if name.endswith("u1") else 10.0
    if name.startswith("T"):
        p = _transport_params(T=5, reward_class=TunedReward1D(50000, -1e3, 3e2),
                              reset_init_condition_func=lambda nx: np.ones(nx) * u0, reset_recirculation_func=lambda nx: g.beta)
        env = pde_control_gym.make("PDEControlGym-TransportPDE1D", **p)
        ctrl = lambda o: np.dot(g.kernel, o.astype(np.float64)) * 1e-2
    else:
        p = _transport_params(T=1, dt=1e-5, dx=5e-3, control_sample_rate=1e-3, reward_class=TunedReward1D(100000, -1e3, 3e2),
                              reset_init_condition_func=lambda nx: np.ones(nx + 1) * u0, reset_recirculation_func=lambda nx: g.beta)
        env = pde_control_gym.make("PDEControlGym-ReactionDiffusionPDE1D", **p)
        m = min(len(g.kernel_row), 200)
        ctrl = lambda o: np.sum(g.kernel_row[:m] * o[:m].astype(np.float64)) * 5e-3
    obs, _ = env.reset()
    total, l2, te, tr, n = 0.0, 0.0, False, False, 0
    while not te and not tr:
        obs, r, te, tr, _ = env.step(ctrl(obs))
        total += float(r)
        l2 += float(np.linalg.norm(obs))
        n += 1
    assert n == len(g.actions)
    pub_total, pub_l2 = KAT_PUBLISHED[name]
    np.testing.assert_allclose(total, pub_total, rtol=1e-5)
    np.testing.assert_allclose(l2, pub_l2, rtol=1e-5)
    np.testing.assert_array_equal(obs, g.last_obs)          # final state bit-identical to the reference's


def test_docs_quickstart_is_bit_reproduced_on_gpu(golden_mixed):
    """docs/source/guide/quickstart.rst:9-70 as written (float64 ``np.ones(nx)`` beta, ``normalize: None``, ``env.step(0)``
    with a Python int) through the drop-in API: every observation equals the reference's, bit for bit."""
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    g = golden_mixed["Q_quick"]
    T, dt, dx, X = 5, 1e-4, 1e-2, 1
    params = {"T": T, "dt": dt, "X": X, "dx": dx, "reward_class": TunedReward1D(int(round(T / dt)), -1e-4, 1e2), "normalize": None,
              "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": lambda state: state,
              "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20,
              "reset_init_condition_func": lambda nx: g.init, "reset_recirculation_func": lambda nx: np.ones(nx),
              "control_sample_rate": 0.1}
    env = pde_control_gym.make("PDEControlGym-TransportPDE1D", **params)
    obs, _ = env.reset()
    np.testing.assert_array_equal(obs, g.obs[0])
    terminate = truncate = False
    total, k = 0.0, 0
    while not truncate and not terminate:
        obs, rewards, terminate, truncate, info = env.step(0)
        total += rewards
        k += 1
        np.testing.assert_array_equal(obs, g.obs[k], err_msg=f"step {k}")
        assert terminate == bool(g.terminate[k - 1]) and truncate == bool(g.truncate[k - 1])
    np.testing.assert_allclose(total, float(np.sum(g.reward[:k])), rtol=1e-6)


@pytest.mark.parametrize("case", ["Q_p_b64_neu_py", "Q_t_b32_neu_np64_norm", "Q_p_b32_neu_py_norm", "Q_t_b64_py_norm"])
def test_dropin_env_follows_the_type_of_the_control_argument(golden_mixed, case):
    """The single-environment classes classify ``control`` like NumPy does (Python float = weak scalar, np.float64, float32
    array) and keep the dtype of the plant parameter: bit-exact against the reference for each combination."""
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    from tests.cases import MIXED_CASES
    kind, kw, action_as, _, _ = MIXED_CASES[case]
    g = golden_mixed[case]
    p = dict(kw, reward_class=TunedReward1D(int(g.reward_args[0]), g.reward_args[1], g.reward_args[2]),
             sensing_noise_func=lambda s: s, reset_init_condition_func=lambda nx: g.init, reset_recirculation_func=lambda nx: g.beta)
    env = pde_control_gym.make("PDEControlGym-ReactionDiffusionPDE1D" if kind == "parabolic" else "PDEControlGym-TransportPDE1D", **p)
    obs, _ = env.reset()
    conv = {"pyfloat": float, "npf64": np.float64, "f32arr": lambda a: np.array([a], dtype=np.float32), "pyint": int}[action_as]
    for i, a in enumerate(g.actions):
        obs, r, te, tr, _ = env.step(conv(a))
        np.testing.assert_array_equal(np.asarray(obs).reshape(-1), g.obs[i + 1], err_msg=f"step {i}")


def test_vecenv_gpu_autoreset_matches_single_envs():
    """PDEVecEnv on the GPU: SB3 semantics with B instances == B single environments stepped one by one."""
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    B = 6
    p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, reward_class=TunedReward1D(400, -1e3, 3e2))
    ics = [np.ones(100, dtype=np.float32) * (1.5 + 0.25 * k) for k in range(64)]
    it = iter(ics)
    p["reset_init_condition_func"] = lambda nx: next(it)
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **p)
    obs = venv.reset()
    singles = []
    for b in range(B):
        q = dict(p)
        q["reset_init_condition_func"] = lambda nx, b=b: ics[b]
        e = pde_control_gym.make("PDEControlGym-TransportPDE1D", **q)
        o, _ = e.reset()
        np.testing.assert_array_equal(o, obs[b])
        singles.append(e)
    rng = np.random.default_rng(0)
    for k in range(14):
        a = rng.uniform(-1, 1, (B, 1)).astype(np.float32)
        obs, rew, dones, infos = venv.step(a)
        for b in range(B):
            o, r, te, tr, _ = singles[b].step(a[b])
            np.testing.assert_allclose(rew[b], r, rtol=1e-6, atol=1e-5)
            if k < 13:
                np.testing.assert_array_equal(obs[b], o)
            else:
                assert te and dones[b]
                np.testing.assert_array_equal(infos[b]["terminal_observation"], o)
                np.testing.assert_array_equal(obs[b], ics[B + b])       # next draws of the reset callback


def test_ns_public_api_on_gpu(golden_ns):
    from pde_control_gym.src import NavierStokes2D, NSReward
    g = golden_ns["N1"]
    Uref = np.zeros((200, 21, 21, 2))
    for k in g.keep:
        Uref[int(k)] = np.stack([g[f"u{int(k)}"], g[f"v{int(k)}"]], -1)
    p = {"T": 0.2, "dt": 1e-3, "X": 1, "dx": 0.05, "Y": 1, "dy": 0.05, "action_dim": 1, "reward_class": NSReward(0.1),
         "normalize": False, "reset_init_condition_func": lambda X: (g.u0.copy(), g.v0.copy(), np.zeros_like(X)),
         "boundary_condition": NS_BC, "U_ref": Uref, "action_ref": 2.0 * np.ones(1000)}
    env = NavierStokes2D(**p)
    env.reset(seed=400)
    for t in range(1, 51):
        obs, r, te, tr, _ = env.step(g.actions[t - 1])
        if t in (1, 2, 50):
            np.testing.assert_array_equal(obs[..., 0], g[f"u{t}"])
            np.testing.assert_array_equal(env.U[t, :, :, 1], g[f"v{t}"])
            np.testing.assert_allclose(r, g.rewards[t - 1], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(env.u, g["u50"])
    np.testing.assert_array_equal(env.u, env._core.u[0].cpu().numpy())     # the host view IS the device state (u, v)
    np.testing.assert_array_equal(env.v, env._core.v[0].cpu().numpy())
    # adjoint-example usage: solve_pressure on caller arrays
    from pde_control_gym.src.environments2d.navier_stokes2D import central_difference
    pr = env.solve_pressure(env.u, env.v, np.zeros((21, 21)))
    assert np.isfinite(central_difference(pr, "x", 0.05)).all()


def test_device_rollout_graph_equals_eager_on_gpu():
    """The hipGraph-captured rollout (policy MLP + fused env step, zero host work per step) reproduces the eager loop
    bit for bit, across an episode boundary (fused auto-reset)."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    from pde_control_gym.src import TunedReward1D
    B, T = 64, 20                                   # nt=401, S=30 -> episodes end every 14 steps
    torch.manual_seed(0)
    pol = torch.nn.Sequential(torch.nn.Linear(100, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1), torch.nn.Tanh()).cuda()
    outs = []
    for graph in (False, True):
        p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, reward_class=TunedReward1D(400, -1e3, 3e2))
        rng = np.random.default_rng(5)
        p["reset_init_condition_func"] = lambda nx: np.ones(nx) * rng.uniform(1, 3)
        venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **p)
        venv.reset_tensor()
        venv.enable_fused_auto_reset()
        ro = DeviceRollout(venv, pol, T, use_graph=graph).run()
        torch.cuda.synchronize()
        outs.append([x.cpu().numpy().copy() for x in (ro.obs, ro.actions, ro.rewards, ro.terminated, ro.truncated)])
        if graph:                                    # a second replay continues from the rollout's last observation
            ro.run()
            torch.cuda.synchronize()
            assert np.isfinite(ro.rewards.cpu().numpy()).all()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert outs[0][3][13].all() and not outs[0][3][12].any()      # every instance terminates at step 14 and restarts


@pytest.mark.parametrize("one_launch", [False, True])
def test_device_rollout_replay_after_plain_steps_keeps_the_state(one_launch):
    """A plain step between two replays flips the engine's double-buffered observation (= state) tensors; the captured graph
    still writes its end state into the buffer it was captured with, so the engine has to be pointed back at it.  run(), step,
    run(), step must equal the eager path bit for bit (round-2 advisor finding: the second run used to hand a stale state on)."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout, FusedMLP
    from pde_control_gym.src import TunedReward1D
    B, T = 48, 6
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(100, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1)).cuda()
    outs = []
    for graph in (False, True):
        p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, reward_class=TunedReward1D(400, -1e3, 3e2))
        rng = np.random.default_rng(7)
        p["reset_init_condition_func"] = lambda nx: np.ones(nx) * rng.uniform(1, 3)
        venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **p)
        venv.reset_tensor()
        venv.enable_fused_auto_reset()
        pol = FusedMLP(net) if one_launch else net
        ro = DeviceRollout(venv, pol, T, use_graph=graph, one_launch=one_launch)
        res = []
        a = torch.full((B,), 0.25, device="cuda")
        for rep in range(3):
            ro.run()
            res += [ro.obs.cpu().numpy().copy(), ro.rewards.cpu().numpy().copy()]
            for k in range(1 + rep):               # an odd and an even number of plain steps between replays
                o, r, te, tr = venv.step_tensor(a)
                res += [o.cpu().numpy().copy(), r.cpu().numpy().copy()]
        outs.append(res)
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


def test_device_sensing_noise_in_the_rollout_graph():
    """sensing_noise_tensor_func (the device-side twin of hyperbolic.py:160-164's hook): the policy sees f(obs) while the plant
    state stays clean.  A deterministic f: graph == eager bit for bit, obs_seen == f(obs), the trajectory is that of a policy
    composed with f on an environment without the hook.  A random f inside the graph: every replay draws new noise."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    from pde_control_gym.src import TunedReward1D
    B, T = 32, 8
    torch.manual_seed(2)
    pol = torch.nn.Sequential(torch.nn.Linear(100, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1), torch.nn.Tanh()).cuda()
    f = lambda o: o * 1.03125 + 0.25

    def venv(noise):
        p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, reward_class=TunedReward1D(400, -1e3, 3e2))
        rng = np.random.default_rng(9)
        p["reset_init_condition_func"] = lambda nx: np.ones(nx) * rng.uniform(1, 3)
        v = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, sensing_noise_tensor_func=noise, **p)
        v.reset_tensor()
        v.enable_fused_auto_reset()
        return v

    outs = []
    for graph in (False, True):
        ro = DeviceRollout(venv(f), pol, T, use_graph=graph).run()
        torch.cuda.synchronize()
        assert not ro.one_launch
        torch.testing.assert_close(ro.obs_seen, f(ro.obs), rtol=0, atol=0)
        outs.append([x.cpu().numpy().copy() for x in (ro.obs, ro.actions, ro.rewards)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    ref = DeviceRollout(venv(None), lambda o: pol(f(o)), T, use_graph=False).run()
    np.testing.assert_array_equal(ref.obs.cpu().numpy(), outs[0][0])
    np.testing.assert_array_equal(ref.actions.cpu().numpy(), outs[0][1])
    # the step_tensor face returns the noisy observation, the state stays clean
    v = venv(f)
    clean = v.core.t["obs"].clone()
    o, *_ = v.step_tensor(torch.zeros(B, device="cuda"))
    torch.testing.assert_close(o, f(v.core.t["obs"]), rtol=0, atol=0)
    assert not torch.equal(v.core.t["obs"], clean)
    # random noise captured in the graph
    ro = DeviceRollout(venv(lambda o: o + 0.01 * torch.randn_like(o)), pol, T, use_graph=True).run()
    n1 = (ro.obs_seen - ro.obs).clone()
    ro.run()
    n2 = ro.obs_seen - ro.obs
    assert torch.isfinite(n1).all() and n1.abs().max() > 0 and not torch.equal(n1, n2)


def test_quickstart_example_runs():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "quickstart.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("quickstart ok"), r.stdout[-2000:]


def test_ppo_example_trains_on_device():
    """examples/train_ppo_device.py: PPO whose rollouts are one hipGraph replay each (FusedMLP actor with exploration noise +
    env-step with fused auto-reset) and whose updates are ordinary torch autograd on the same module; a short run has to
    improve the episode return of the unstable reaction-diffusion plant."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_ppo_device.py")
    spec = importlib.util.spec_from_file_location("train_ppo_device", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.main(iterations=14, B=1024, quiet=True)
    first = hist[0]["episode_return"]
    last = sum(h["episode_return"] for h in hist[-3:]) / 3
    assert last > first + 15.0, (first, last)
    assert hist[-1]["mean_norm"] < hist[0]["mean_norm"]
    # the same with SB3's 256-256 width: the rollout stays ONE launch (cooperative MFMA evaluation inside the kernel) and learns
    hist = mod.main(iterations=10, B=512, quiet=True, hidden=256)
    assert sum(h["episode_return"] for h in hist[-3:]) / 3 > hist[0]["episode_return"] + 10.0, hist


@pytest.mark.parametrize("family", ["burgers", "traffic"])
def test_device_rollout_graph_equals_eager_other_families(family):
    """Graph replay == eager loop for the remaining environment families (the Burgers extension on the transport kernel; the
    traffic engine, whose action slot is rebound by every step)."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    from pde_control_gym.src import TunedReward1D
    torch.manual_seed(0)
    outs = []
    for graph in (False, True):
        if family == "burgers":
            p = _transport_params(T=0.0400, dt=1e-4, control_sample_rate=30e-4, reward_class=TunedReward1D(400, -1e3, 3e2))
            rng = np.random.default_rng(5)
            p["reset_init_condition_func"] = lambda nx: np.ones(nx) * rng.uniform(0.2, 0.6)
            venv = pde_control_gym.make_vec("PDEControlGym-BurgersPDE1D", num_envs=48, **p)
            venv.reset_tensor()
            venv.enable_fused_auto_reset()
            pol = torch.nn.Sequential(torch.nn.Linear(100, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1), torch.nn.Tanh()).cuda()
            lo, hi = -1.0, 1.0
        else:
            import random
            from pde_control_gym.src import TrafficARZReward
            random.seed(0)
            venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=48, reward_class=TrafficARZReward(),
                                            simulation_type="both", limit_pde_state_size=True, control_freq=2, T=240, dt=0.25, X=500, dx=10)
            venv.reset_tensor()
            torch.manual_seed(2)
            lin = torch.nn.Linear(102, 2).double().cuda()
            pol = lambda o: 4.6 + 0.5 * torch.tanh(lin(o))             # noqa: E731
            lo, hi = 3.0, 6.0
        torch.manual_seed(1)
        if family == "burgers":
            for q in pol.parameters():
                torch.nn.init.normal_(q, std=0.2)
        ro = DeviceRollout(venv, pol, 16, use_graph=graph, action_low=lo, action_high=hi).run()
        ro.run()
        torch.cuda.synchronize()
        outs.append([x.cpu().numpy().copy() for x in (ro.obs, ro.actions, ro.rewards, ro.terminated, ro.truncated)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(outs[0][2]).all()


def test_vecenv_host_arrays_are_never_overwritten_while_referenced():
    """PDEVecEnv.step hands out views of pinned staging buffers recycled by reference count (advisor finding r3: rotating buffers
    silently overwrote results a caller kept).  Dropped results: the pool stays at two or three buffers.  Kept results -- also
    through a slice -- keep their values for good, beyond ``host_buffers`` as plain copies."""
    import pde_control_gym
    B = 8
    p = _transport_params(T=0.5, dt=1e-4, control_sample_rate=0.01)
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **p)
    venv.host_buffers = 6
    venv.reset()
    a = np.full((B, 1), 0.5, np.float32)
    for _ in range(12):
        obs, rew, dones, infos = venv.step(a)
    pool = [v for k, v in venv._pins.items() if k[0] == "out" and k[1] == 0][0]
    assert len(pool) <= 3
    kept, want, rows = [], [], []
    for k in range(10):                                   # more than host_buffers
        obs, rew, dones, infos = venv.step(a * (1 + 0.1 * k))
        kept.append(obs)
        want.append(obs.copy())
        rows.append((obs[3], obs[3].copy()))
        del obs
    assert len({id(x) for x in kept}) == 10
    for x, y in zip(kept, want):
        np.testing.assert_array_equal(x, y)
    del kept
    for _ in range(8):                                    # the row views still pin their base arrays
        venv.step(a)
    for view, copy in rows:
        np.testing.assert_array_equal(view, copy)


@pytest.mark.gpu
def test_vecenv_copy_outputs_true_hands_out_plain_copies_and_non_cpython_rule():
    """``copy_outputs=True`` is honoured -- every result is a plain NumPy copy (no pinned pool is kept per output), with
    the same values as the recycling default; and the recycling path is only taken where sys.getrefcount is exact (CPython + GIL)."""
    import pde_control_gym
    from pde_control_gym import vector as V
    assert V._REFCOUNT_IS_EXACT == (sys.implementation.name == "cpython" and getattr(sys, "_is_gil_enabled", lambda: True)())
    B = 6
    p = _transport_params(T=0.5, dt=1e-4, control_sample_rate=0.01)
    a = np.full((B, 1), 0.25, np.float32)
    res = {}
    for mode in (None, True):
        venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, copy_outputs=mode, **p)
        venv.reset()
        outs = [venv.step(a * (1 + 0.2 * k)) for k in range(6)]
        res[mode] = [(o.copy(), r.copy(), d.copy()) for o, r, d, _ in outs]
        pools = [v for k, v in venv._pins.items() if k[0] == "out"]
        if mode is True:
            assert all(len(v) == 0 for v in pools)                       # nothing recycled: staged through the scratch buffer, copied out
            assert len({id(o) for o, _, _, _ in outs}) == 6
            assert not any(np.shares_memory(outs[0][0], o) for o, _, _, _ in outs[1:])
        else:
            assert any(len(v) > 0 for v in pools)
    for x, y in zip(res[None], res[True]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(u, v)


@pytest.mark.gpu
def test_hbm_probe_copies_and_reports_plausible_rates():
    """tools/hbm_probe.hip (the yardstick bench.py prints beside the 8 TB/s specification): the copy kernel really copies, read and
    fill run, and the rates are those of an HBM3E part, not of a mis-timed launch (256 MiB: above 1 TB/s, below the specification
    plus what the 256 MiB memory-side cache can add)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hbm_probe as hp
    r = hp.measure("cuda", sizes_mib=(256,), grids=(4096, 16384))
    for k in ("copy_GBps", "read_GBps", "fill_GBps"):
        assert 1000.0 < r[k] < 20000.0, (k, r[k])
    n = 1 << 22
    src = torch.randn(n, device="cuda")
    dst = torch.zeros_like(src)
    for nt in (0, 1):
        dst.zero_()
        assert hp.lib().pdegym_probe_hbm(0, dst.data_ptr(), src.data_ptr(), n * 4, 333, nt, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(dst, src)
    assert hp.lib().pdegym_probe_hbm(0, dst.data_ptr(), src.data_ptr(), n * 4 + 4, 333, 0, None) == -1      # not a multiple of 16 bytes


def test_single_env_bench_block_runs_and_reports_every_shape():
    """bench.py's single_env block (bench_single.py): every shape is measured on both sides, the shipped example shapes are faster on the
    GPU face than in NumPy, and the crossover is a small positive number of sub-steps."""
    import bench_single
    out = bench_single.single_env_block(torch.device("cuda", 0), seconds=0.08)
    for name in ("transport_c1", "parabolic_example", "transport_s1", "ns2d_example", "traffic_example", "tumor_example"):
        assert "error" not in out[name], out[name]
        assert out[name]["gpu"]["us_per_step"] > 0 and out[name]["numpy"]["us_per_step"] > 0
    assert out["transport_c1"]["speedup"] > 5 and out["parabolic_example"]["speedup"] > 3 and out["ns2d_example"]["speedup"] > 5
    assert 0 < out["crossover"]["substeps_above_which_gpu_wins"] < 50


@pytest.mark.parametrize("kind,history", [("transport", True), ("transport", False), ("parabolic", True), ("parabolic", False)])
def test_single_env_host_pack_mirrors_the_device_state(kind, history):
    """The batch-of-one face keeps time_index on the host (min(t + S, nt - 1), hyperbolic.py:140) and reads observation / reward / flags
    from pinned host views the kernel wrote: after every step they equal what the device holds, through the end of the episode and
    past it (post-terminal calls do nothing)."""
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    nx = 40
    n = nx + (kind == "parabolic")
    dt, dx = (1e-4, 1.0 / nx) if kind == "parabolic" else (5e-3, 1.0 / nx)
    S, T = 7, 45 * dt
    env = pde_control_gym.make("PDEControlGym-TransportPDE1D" if kind == "transport" else "PDEControlGym-ReactionDiffusionPDE1D",
                               device="cuda", record_history=history, T=T, dt=dt, X=1, dx=dx, control_sample_rate=S * dt,
                               reward_class=TunedReward1D(int(round(T / dt)), -1e3, 3e2), normalize=True, sensing_loc="full",
                               control_type="Dirchilet", sensing_type=None, sensing_noise_func=lambda s: s, limit_pde_state_size=True,
                               max_state_value=1e10, max_control_value=20, reset_init_condition_func=lambda nx_: np.linspace(1, 2, n),
                               reset_recirculation_func=lambda nx_: np.ones(n, dtype=np.float32)).unwrapped
    env.reset()
    core = env._core
    for k in range(9):                  # 45 / 7: the episode ends inside step 7; steps 8 and 9 are post-terminal
        obs, r, te, tr, _ = env.step(np.array([0.1 * k - 0.3], dtype=np.float32))
        assert env.time_index == int(core.time_index.cpu()[0]) == min(7 * (k + 1), core.nt - 1)
        np.testing.assert_array_equal(obs, core.u.cpu().numpy()[0])
        assert te == bool(core.t["terminated"][0]) == (env.time_index >= core.nt - 1) and tr == bool(core.t["truncated"][0])
        if history:
            np.testing.assert_array_equal(env.u[env.time_index], obs)
    assert te


def test_integration_md_binding_stub_reproduces_the_reference_golden(golden_transport):
    """INTEGRATION.md section 2 as code: the ctypes stub a maintainer of the reference would put in place of hyperbolic.py:137-169 -- raw
    C-ABI structures, device tensors for the plant state, ONE pinned host allocation for the command and the results (the kernel reads
    and writes it in place), one launch + one synchronisation per step -- reproduces the reference-generated golden H1 (nx = 100,
    S = 1000) bit for bit, without any of this repository's engine classes."""
    import ctypes as C
    from pdecontrolgym_amd import _native as N
    from tests.cases import TRANSPORT_CASES
    lib = N.load()
    kw, g = TRANSPORT_CASES["H1"], golden_transport["H1"]
    n, nt, S = 100, int(round(kw["T"] / kw["dt"]) + 1), int(round(kw["control_sample_rate"] / kw["dt"]))
    P = N.Params1D()
    P.n, P.nt, P.substeps, P.control_type, P.normalize, P.sensing, P.limit_state = n, nt, S, 0, 0, N.SENSE_FULL, 1
    P.reward_kind, P.reward_nt = N.REWARD_TUNED1D, int(g.reward_args[0])
    P.truncate_penalty, P.terminate_reward = float(g.reward_args[1]), float(g.reward_args[2])
    P.dt, P.dx, P.F, P.max_control, P.max_state = kw["dt"], kw["dx"], kw["dt"] / kw["dx"] ** 2, 20, 1e10
    P.rdx = 1.0 / float(C.c_float(kw["dx"]).value)
    P.dt64, P.dx64, P.max_control64 = kw["dt"], kw["dx"], 20
    dev = torch.device("cuda")
    u = torch.zeros(1, n, device=dev)
    beta = torch.as_tensor(np.asarray(g.beta, dtype=np.float32), device=dev)
    t_idx = torch.zeros(1, dtype=torch.int32, device=dev)
    bsum = torch.zeros(1, dtype=torch.float64, device=dev)
    ring = torch.zeros(1, N.RING, device=dev)
    norm_back = torch.zeros(1, device=dev)
    hist = torch.zeros(1, nt, n, device=dev)
    pack = torch.zeros(8 + 4 * n + 4 + 4 + 2, dtype=torch.uint8, pin_memory=True)
    B = N.Bufs1D()
    B.u, B.beta, B.beta_stride = u.data_ptr(), beta.data_ptr(), 0
    B.time_index, B.bsum, B.ring, B.norm_back, B.history = t_idx.data_ptr(), bsum.data_ptr(), ring.data_ptr(), norm_back.data_ptr(), hist.data_ptr()
    base = pack.data_ptr()
    B.action, B.obs = base, base + 8
    B.reward, B.norm_now = base + 8 + 4 * n, base + 12 + 4 * n
    B.terminated, B.truncated = base + 16 + 4 * n, base + 17 + 4 * n
    host = pack.numpy()
    a32, obs_h, rew_h, flags_h = host[:4].view(np.float32), host[8:8 + 4 * n].view(np.float32), host[8 + 4 * n:12 + 4 * n].view(np.float32), host[16 + 4 * n:18 + 4 * n]
    init = torch.as_tensor(np.asarray(g.init, dtype=np.float32)[None], device=dev)
    stream = torch.cuda.current_stream()
    N.check(lib.pdegym_reset1d_masked(C.byref(P), C.byref(B), init.data_ptr(), None, 1, stream.cuda_stream), "reset")
    stream.synchronize()
    np.testing.assert_array_equal(obs_h, g.obs[0])
    t = 0
    for i, a in enumerate(g.actions):
        a32[0] = a
        P.action_kind = N.ACTION_F32
        N.check(lib.pdegym_transport_step(C.byref(P), C.byref(B), 1, stream.cuda_stream), "step")
        stream.synchronize()
        t = min(t + S, nt - 1)
        np.testing.assert_array_equal(obs_h, g.rows[i], err_msg=f"row of step {i}")
        assert t == int(g.time_index[i]) == int(t_idx.cpu()[0])
        assert bool(flags_h[0]) == bool(g.terminate[i]) and bool(flags_h[1]) == bool(g.truncate[i])
        if np.isfinite(g.reward[i]):
            np.testing.assert_allclose(rew_h[0], g.reward[i], rtol=1e-6, atol=1e-6 * max(1.0, float(np.linalg.norm(g.rows[i]))))
    np.testing.assert_array_equal(hist[0, t].cpu().numpy(), g.rows[-1])          # the trajectory (env.u) on the device
