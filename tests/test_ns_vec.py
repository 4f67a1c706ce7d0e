"""NavierStokes2D on the batched face: fused auto-reset (inside the C-ABI call) and the on-device rollout loop -- the training
loop pieces of SURVEY.md section 8f rank 1 for the environment the reference trains PPO / SAC on
(examples/NavierStokes/NS2Dppo.py:29-66).  Runs on the CPU double here and on the HIP backend under -m gpu."""
import numpy as np
import pytest
import torch

from tests.cases import NS_BC
from tests.fake_backend import FakeBackend

BACKENDS = [pytest.param("double", id="cpu-double"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _bk(kind):
    return dict(device="cpu", backend=FakeBackend()) if kind == "double" else dict(device="cuda")


def _params(n, nt, K, rng, dtype):
    from pde_control_gym.src import NSReward
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    return {"T": nt * dt, "dt": dt, "X": 1, "dx": dx, "Y": 1, "dy": dx, "action_dim": 1, "reward_class": NSReward(0.1),
            "normalize": False, "boundary_condition": NS_BC, "U_ref": rng.uniform(-1, 1, (nt, n, n, 2)), "action_ref": 2.0 * np.ones(nt),
            "maximum_pressure_iteration": K, "dtype": dtype}


@pytest.mark.parametrize("bk", BACKENDS)
@pytest.mark.parametrize("n,dtype", [(21, "float64"), (128, "float32"), (256, "float32")])
def test_ns_fused_auto_reset_equals_manual_resets(bk, n, dtype):
    """Episodes of 4 steps; with pools of 3B rows the k-th restart of instance b starts from row (b + k B) mod 3B.  Observations,
    rewards, flags and terminal observations equal those of a second engine that is reset by hand."""
    import pde_control_gym
    if bk == "double" and n > 21:
        pytest.skip("the NumPy double is slow on big grids; the HIP run covers them")
    rng = np.random.default_rng(n)
    B, nt, K = 3, 6, 5
    p = _params(n, nt, K, rng, dtype)
    pools = [rng.uniform(-1, 1, (3 * B, n, n)) for _ in range(3)]
    it = iter(range(10 ** 6))
    p["reset_init_condition_func"] = lambda X: tuple(pool[next(it) % B] for pool in pools)      # only used by reset()
    venv = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, **_bk(bk), **p)
    ref = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, **_bk(bk), **p)
    first = [pool[:B] for pool in pools]
    venv.core.reset(*first)
    ref.core.reset(*first)
    venv.enable_fused_auto_reset(tuple(pools))
    restarts = np.zeros(B, dtype=int)
    for step in range(11):
        a = torch.as_tensor(rng.uniform(2, 4, (B, 1)), dtype=venv.core.dtype, device=venv.device)
        obs, r, te, tr = venv.step_tensor(a)
        o2, r2, te2, _ = ref.step_tensor(a)
        torch.testing.assert_close(r, r2, rtol=0, atol=0)
        assert torch.equal(te, te2)
        done = te2.cpu().numpy().astype(bool)
        if done.any():
            assert torch.equal(venv.core.t["final_obs"][te2.bool()], o2[te2.bool()])
            rows = (np.arange(B) + restarts * B) % (3 * B)
            restarts[done] += 1
            mask = torch.as_tensor(done.astype(np.uint8), device=venv.device)
            o2 = ref.core.reset(*[pool[rows] for pool in pools], mask=mask)
        assert torch.equal(obs, o2), f"step {step}"
        assert torch.equal(venv.core.p, ref.core.p)
        assert torch.equal(venv.core.time_index, ref.core.time_index)
    assert restarts.min() >= 2 and (venv.core.t["reset_count"].cpu().numpy() == restarts).all()


@pytest.mark.parametrize("bk", BACKENDS)
def test_ns_device_rollout_equals_stepping_by_hand(bk):
    """DeviceRollout on NavierStokes2D (policy forward + env step per slot, the observation buffer doubling as the state;
    on the GPU the whole rollout is one hipGraph): same observations / rewards as calling step_tensor in a Python loop, across
    an episode boundary with the fused auto-reset, and the engine can be stepped normally afterwards."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    n, B, nt, K, T = 21, 4, 7, 6, 9
    outs = []
    torch.manual_seed(0)
    dev = "cpu" if bk == "double" else "cuda"
    lin = torch.nn.Linear(n * n * 2, 1).double().to(dev)
    policy = lambda o: 3.0 + torch.tanh(lin(o.reshape(o.shape[0], -1)))
    for mode in ("hand", "rollout", "rollout"):
        rng = np.random.default_rng(3)
        p = _params(n, nt, K, rng, "float64")
        pools = [rng.uniform(-1, 1, (2 * B, n, n)) for _ in range(3)]
        p["reset_init_condition_func"] = lambda X: (pools[0][0], pools[1][0], pools[2][0])
        venv = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, **_bk(bk), **p)
        venv.core.reset(*[pool[:B] for pool in pools])
        venv.enable_fused_auto_reset(tuple(pools))
        if mode == "hand":
            obs = [venv.core.t["obs"].clone()]
            rews, terms = [], []
            for t in range(T):
                with torch.no_grad():
                    a = policy(obs[-1]).reshape(B, 1).clamp(2.0, 4.0)
                o, r, te, _ = venv.step_tensor(a)
                obs.append(o.clone()); rews.append(r.clone()); terms.append(te.clone())
            outs.append((torch.stack(obs), torch.stack(rews), torch.stack(terms)))
            nxt = venv.step_tensor(torch.full((B, 1), 2.5, dtype=torch.float64, device=dev))
            outs.append(nxt[0].clone())
        else:
            ro = DeviceRollout(venv, policy, T, use_graph=(len(outs) > 3), action_low=2.0, action_high=4.0).run()
            outs.append((ro.obs.clone(), ro.rewards.clone(), ro.terminated.clone()))
            nxt = venv.step_tensor(torch.full((B, 1), 2.5, dtype=torch.float64, device=dev))
            outs.append(nxt[0].clone())
    (o0, r0, t0), n0, (o1, r1, t1), n1, (o2, r2, t2), n2 = outs
    for o, r, t, nx in ((o1, r1, t1, n1), (o2, r2, t2, n2)):
        assert torch.equal(o0, o) and torch.equal(r0, r) and torch.equal(t0, t)
        assert torch.equal(n0, nx)                 # the engine continues from the rollout's last state
    assert t0.sum() > 0                            # an episode boundary was crossed


@pytest.mark.gpu
def test_ns_device_rollout_with_fused_policy():
    """FusedMLP on the flattened float64 Navier-Stokes observation (21 x 21 x 2 = 882 inputs, rounded to float32 as they are
    read; lid action in [2, 4] through the fused clamp): same rollout as the wrapped torch module within the float32 agreement
    of the forward passes, and the graph replay equals the eager run bit for bit."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout, FusedMLP
    n, B, nt, K, T = 21, 6, 7, 6, 9
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(n * n * 2, 32), torch.nn.Tanh(), torch.nn.Linear(32, 1)).cuda()
    with torch.no_grad():
        net[-1].bias.fill_(3.0)
    runs = {}
    for name, pol, graph in (("torch", lambda o: net(o.reshape(o.shape[0], -1).float()).double(), False),
                             ("fused", FusedMLP(net), False), ("fused_graph", FusedMLP(net), True)):
        rng = np.random.default_rng(3)
        p = _params(n, nt, K, rng, "float64")
        pools = [rng.uniform(-1, 1, (2 * B, n, n)) for _ in range(3)]
        p["reset_init_condition_func"] = lambda X: (pools[0][0], pools[1][0], pools[2][0])
        venv = pde_control_gym.make_vec("PDEControlGym-NavierStokes2D", num_envs=B, **p)
        venv.core.reset(*[pool[:B] for pool in pools])
        venv.enable_fused_auto_reset(tuple(pools))
        ro = DeviceRollout(venv, pol, T, use_graph=graph, action_low=2.0, action_high=4.0).run()
        torch.cuda.synchronize()
        runs[name] = [x.cpu().numpy().copy() for x in (ro.actions, ro.obs, ro.rewards)]
    assert runs["fused"][0].min() >= 2.0 and runs["fused"][0].max() <= 4.0 and runs["fused"][0].std() > 0
    for got, want in zip(runs["fused"], runs["torch"]):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    for got, want in zip(runs["fused_graph"], runs["fused"]):
        np.testing.assert_array_equal(got, want)
