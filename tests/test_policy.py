"""FusedMLP (pdegym_mlp_forward): one-launch forward pass of an SB3-style MLP policy.

GPU tests compare the HIP kernel with the torch module it wraps (float32 rounding tolerance: the kernel sums k in ascending
order with FMAs, a BLAS GEMM does not); CPU tests drive the host wrapper through the oracle-backed double."""
import numpy as np
import pytest
import torch

from pdecontrolgym_amd import _native as N
from pdecontrolgym_amd.policy import FusedMLP


def _mlp(sizes, acts, bias=True, seed=0):
    torch.manual_seed(seed)
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(torch.nn.Linear(sizes[i], sizes[i + 1], bias=bias))
        if acts[i] == "tanh":
            mods.append(torch.nn.Tanh())
        elif acts[i] == "relu":
            mods.append(torch.nn.ReLU())
    return torch.nn.Sequential(*mods)


SHAPES = [
    ([257, 64, 64, 1], ["tanh", "tanh", "tanh"]),          # SB3 MlpPolicy on the C2 observation, squashed output
    ([513, 64, 64, 1], ["tanh", "tanh", None]),            # C3 observation, linear head
    ([101, 128, 2], ["relu", None]),
    ([7, 256, 200, 33, 3], ["tanh", "relu", "tanh", None]),  # four layers, ragged widths
    ([1, 1], [None]),
    ([300, 65, 1], ["tanh", "tanh"]),                      # width just above one neuron tile
]


def test_struct_layout_matches_header():
    import ctypes as C
    assert C.sizeof(N.MlpLayer) == 32 and C.sizeof(N.Mlp) == 40 + 4 * 32
    assert N.Mlp.layer.offset == 40 and N.Mlp.x_f64.offset == 16 and N.Mlp.noise.offset == 24 and N.MlpLayer.in_dim.offset == 16


def test_rejects_unsupported_modules():
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Sigmoid()), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(4, 4)), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Linear(4, 300)), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(9, 1)), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(*[torch.nn.Linear(4, 4) for _ in range(5)]), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Linear(4, 4)).double(), backend=object())
    with pytest.raises(ValueError):
        FusedMLP(torch.nn.Sequential(torch.nn.Linear(4, 1)), clamp=(1.0, -1.0), backend=object())


@pytest.mark.parametrize("sizes,acts", SHAPES[:4])
def test_host_wrapper_with_cpu_double(sizes, acts):
    from tests.fake_backend import FakeBackend
    net = _mlp(sizes, acts)
    pol = FusedMLP(net, clamp=(-1.0, 1.0), backend=FakeBackend())
    x = torch.randn(37, sizes[0])
    with torch.no_grad():
        want = net(x).clamp(-1, 1)
    got = pol(x)
    assert got.shape == (37, sizes[-1])
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
    out = torch.zeros(37, sizes[-1])
    assert pol.forward_into(x, out, clamp=None) is out
    with torch.no_grad():
        np.testing.assert_allclose(out.numpy(), net(x).numpy(), rtol=2e-5, atol=2e-6)
    with pytest.raises(ValueError):
        pol(torch.randn(5, sizes[0] + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,acts", SHAPES)
@pytest.mark.parametrize("B", [1, 16, 37, 4096])
def test_fused_mlp_matches_torch_on_gpu(sizes, acts, B):
    net = _mlp(sizes, acts, seed=B).cuda()
    pol = FusedMLP(net)
    x = torch.randn(B, sizes[0], device="cuda") * 2
    with torch.no_grad():
        want = net(x)
    got = pol(x)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=4e-6)
    # clamp fused, no bias, rows taken from a larger buffer (row stride > row length), output into a [B] view
    net2 = _mlp(sizes[:-1] + [1], acts, bias=False, seed=B + 1).cuda()
    pol2 = FusedMLP(net2, clamp=(-0.25, 0.5))
    big = torch.randn(B, sizes[0] + 5, device="cuda")
    out = torch.full((B,), 7.0, device="cuda")
    pol2.forward_into(big[:, :sizes[0]], out)
    with torch.no_grad():
        want2 = net2(big[:, :sizes[0]]).clamp(-0.25, 0.5).reshape(B)
    np.testing.assert_allclose(out.cpu().numpy(), want2.cpu().numpy(), rtol=2e-5, atol=4e-6)


@pytest.mark.gpu
def test_fused_mlp_sees_in_place_parameter_updates_and_rejects_bad_calls():
    net = _mlp([33, 64, 1], ["tanh", None]).cuda()
    pol = FusedMLP(net)
    x = torch.randn(50, 33, device="cuda")
    a = pol(x).clone()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
        want = net(x)
    b = pol(x)
    assert not torch.allclose(a, b)
    np.testing.assert_allclose(b.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=4e-6)
    with pytest.raises(N.NativeError):
        pol(x.half())
    with pytest.raises(N.NativeError):
        pol(x.cpu())
    # float64 observations / actions (traffic, tumour, float64 Navier-Stokes): rounded on the way in, widened on the way out
    xd = x.double() * (1 + 1e-12)
    got = pol(xd)
    assert got.dtype == torch.float64
    with torch.no_grad():
        want64 = net(xd.float()).double()
    np.testing.assert_allclose(got.cpu().numpy(), want64.cpu().numpy(), rtol=2e-5, atol=4e-6)
    out32 = torch.zeros(50, 1, device="cuda")
    pol.forward_into(xd, out32)
    np.testing.assert_array_equal(out32.cpu().numpy(), got.float().cpu().numpy())


@pytest.mark.gpu
def test_fused_mlp_exploration_noise_before_clamp():
    net = _mlp([40, 64, 2], ["tanh", None]).cuda()
    pol = FusedMLP(net, clamp=(-0.5, 0.5))
    x = torch.randn(70, 40, device="cuda")
    nz = torch.randn(70, 2, device="cuda") * 0.3
    out = torch.zeros(70, 2, device="cuda")
    pol.forward_into(x, out, noise=nz)
    with torch.no_grad():
        want = (net(x) + nz).clamp(-0.5, 0.5)
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=4e-6)
    with pytest.raises(ValueError):
        pol.forward_into(x, out, noise=nz.double())


@pytest.mark.gpu
def test_c_abi_validation_of_mlp_descriptor():
    import ctypes as C
    lib = N.load()
    w = torch.zeros(1, 4, 4, device="cuda")     # blocked transpose [ceil(in_dim / 4), out_dim, 4]
    net = N.Mlp()
    net.n_layers = 1
    net.layer[0].w, net.layer[0].in_dim, net.layer[0].out_dim = w.data_ptr(), 4, 4
    x = torch.zeros(2, 4, device="cuda")
    y = torch.zeros(2, 4, device="cuda")
    call = lambda: lib.pdegym_mlp_forward(C.byref(net), x.data_ptr(), 4, y.data_ptr(), 4, 2, None)
    assert call() == 0
    net.n_layers = 5
    assert call() == -2
    net.n_layers = 1
    net.layer[0].out_dim = 257
    assert call() == -2
    net.layer[0].out_dim = 4
    net.layer[0].act = 9
    assert call() == -2
    net.layer[0].act = 0
    net.clamp, net.lo, net.hi = 1, 1.0, -1.0
    assert call() == -2
    net.clamp = 0
    net.layer[0].w = None
    assert call() == -3
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_device_rollout_with_fused_policy_equals_torch_policy():
    """The rollout driven by FusedMLP (policy + clamp + store in one launch) follows the same trajectory as the torch module
    within the float32 agreement of the two forward passes, in eager mode and replayed from a hipGraph."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout, FusedMLP as Exported
    from pde_control_gym.src import TunedReward1D
    assert Exported is FusedMLP
    B, T, nx = 64, 6, 64
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    beta = np.full(nx + 1, 3.0, np.float32)
    p = {"T": 40 * 5 * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(40 * 5, -1e3, 3e2), "normalize": True,
         "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
         "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 5, "control_sample_rate": 5 * dt,
         "batched_reset_func": lambda idx, nx_: (np.random.default_rng(3).uniform(1, 2, (len(idx), 1)).astype(np.float32)
                                                 * np.ones((1, nx_ + 1), np.float32), np.tile(beta, (len(idx), 1)))}
    net = _mlp([nx + 1, 64, 64, 1], ["tanh", "tanh", "tanh"], seed=5).cuda()
    runs = {}
    for name, pol, graph in (("torch", net, False), ("fused", FusedMLP(net), False), ("fused_graph", FusedMLP(net), True)):
        venv = pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, **p)
        venv.reset_tensor()
        venv.enable_fused_auto_reset()
        ro = DeviceRollout(venv, pol, T, use_graph=graph).run()
        torch.cuda.synchronize()
        runs[name] = (ro.actions.cpu().numpy().copy(), ro.obs.cpu().numpy().copy(), ro.rewards.cpu().numpy().copy())
    for name in ("fused", "fused_graph"):
        for got, want in zip(runs[name], runs["torch"]):
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    for got, want in zip(runs["fused_graph"], runs["fused"]):
        np.testing.assert_array_equal(got, want)


def _rd_env(B, nx, S, horizon, kind="PDEControlGym-ReactionDiffusionPDE1D", seed=3):
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if "Reaction" in kind else 0.5 * dx
    nodes = nx + 1 if "Reaction" in kind else nx
    beta = np.full(nodes, 3.0, np.float32)
    p = {"T": horizon * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(horizon * S, -1e3, 3e2), "normalize": True,
         "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
         "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 5, "control_sample_rate": S * dt,
         "batched_reset_func": lambda idx, nx_: (np.random.default_rng(seed).uniform(1, 2, (len(idx), 1)).astype(np.float32)
                                                 * np.ones((1, nodes), np.float32), np.tile(beta, (len(idx), 1)))}
    venv = pde_control_gym.make_vec(kind, num_envs=B, **p)
    venv.reset_tensor()
    venv.enable_fused_auto_reset()
    return venv


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nx,S,B,sizes,acts", [
    ("PDEControlGym-ReactionDiffusionPDE1D", 256, 20, 100, [257, 64, 64, 1], ["tanh", "tanh", "tanh"]),
    ("PDEControlGym-ReactionDiffusionPDE1D", 64, 5, 37, [65, 48, 1], ["relu", None]),
    ("PDEControlGym-TransportPDE1D", 100, 10, 16, [100, 64, 33, 20, 1], ["tanh", "relu", "tanh", "tanh"]),
    ("PDEControlGym-TransportPDE1D", 512, 4, 5, [512, 32, 64, 1], ["tanh", "tanh", None]),
    ("PDEControlGym-TransportPDE1D", 30, 3, 200, [30, 1], [None]),
    # layers of more than 64 units: the cooperative MFMA evaluation, bit-identical to the two-launch path
    ("PDEControlGym-ReactionDiffusionPDE1D", 256, 20, 100, [257, 256, 256, 1], ["relu", "relu", "tanh"]),
    ("PDEControlGym-ReactionDiffusionPDE1D", 64, 5, 37, [65, 100, 1], ["tanh", None]),
    ("PDEControlGym-TransportPDE1D", 100, 10, 16, [100, 64, 130, 20, 1], ["tanh", "relu", "tanh", "tanh"]),
    ("PDEControlGym-TransportPDE1D", 512, 4, 5, [512, 256, 17, 1], ["tanh", "tanh", None]),
])
def test_one_launch_rollout_with_policy_inside_equals_two_launches_per_step(kind, nx, S, B, sizes, acts):
    """DeviceRollout(one_launch=True): policy + env-step + auto-reset of all T steps in ONE kernel (pdegym_*_rollout with a
    policy) against the policy launch + step launch per env-step.  The environment arithmetic is identical; the two forward
    passes differ in the order of the additions inside groups of 16 inputs (MFMA there, one fma chain here), so commands
    agree to float32 rounding and the trajectories follow (tolerances below); episode ends and restarts coincide."""
    from pde_control_gym import DeviceRollout
    T = 9
    net = _mlp(sizes, acts, seed=11).cuda()
    with torch.no_grad():
        net[0].weight.mul_(0.3)
    runs = {}
    for mode in (False, True):
        venv = _rd_env(B, nx, S, horizon=4, kind=kind)
        ro = DeviceRollout(venv, FusedMLP(net), T, action_low=-2.0, action_high=2.0, action_noise=True, one_launch=mode)
        assert ro.one_launch == mode
        ro.action_noise.copy_(torch.randn(T, B, generator=torch.Generator().manual_seed(1)).mul(0.3).cuda())
        ro.run()
        torch.cuda.synchronize()
        runs[mode] = {k: getattr(ro, k).cpu().numpy().copy() for k in ("actions", "obs", "rewards", "terminated", "truncated")}
        runs[mode]["time_index"] = venv.core.t["time_index"].cpu().numpy().copy()
        runs[mode]["cur"] = venv.core.t["obs"].cpu().numpy().copy()
        ro.run()                                   # the graph replays from the engine's new state
        torch.cuda.synchronize()
        runs[mode]["obs2"] = ro.obs.cpu().numpy().copy()
    a, b = runs[True], runs[False]
    if max(sizes[1:]) > 64:       # the wide evaluation IS pdegym_mlp_forward's reduction: everything equal bit for bit
        for k in ("actions", "obs", "rewards", "cur", "obs2", "terminated", "truncated", "time_index"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_allclose(a["actions"][0], b["actions"][0], rtol=2e-5, atol=4e-6)      # same input: forward passes only
    # ... and against the torch module itself on the first observation slot (noise added before the clamp)
    with torch.no_grad():
        nz0 = torch.randn(T, B, generator=torch.Generator().manual_seed(1)).mul(0.3)[0].cuda()
        want0 = (net(torch.tensor(a["obs"][0]).cuda()).reshape(B) + nz0).clamp(-2.0, 2.0).cpu().numpy()
    np.testing.assert_allclose(a["actions"][0], want0, rtol=2e-5, atol=4e-6)
    for k in ("actions", "obs", "rewards", "cur", "obs2"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ("terminated", "truncated", "time_index"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(a["cur"], a["obs"][T])
    np.testing.assert_array_equal(a["obs2"][0], a["cur"])
    assert a["terminated"].sum() > 0
    assert (np.abs(a["actions"]) == 2.0).sum() < a["actions"].size       # the clamp binds in places at most


@pytest.mark.gpu
@pytest.mark.parametrize("kind,control,loc,stype,nx,S,B,sizes,acts", [
    ("PDEControlGym-ReactionDiffusionPDE1D", "Neumann", "full", None, 128, 10, 50, [129, 64, 64, 1], ["tanh", "tanh", "tanh"]),
    ("PDEControlGym-TransportPDE1D", "Neumann", "full", None, 100, 10, 16, [100, 48, 1], ["relu", None]),
    ("PDEControlGym-TransportPDE1D", "Dirchilet", "collocated", None, 100, 10, 37, [1, 32, 32, 1], ["tanh", "tanh", None]),
    ("PDEControlGym-ReactionDiffusionPDE1D", "Neumann", "collocated", None, 64, 5, 20, [1, 16, 1], ["tanh", None]),
    ("PDEControlGym-ReactionDiffusionPDE1D", "Dirchilet", "full", None, 64, 5, 20, [65, 64, 1], ["tanh", None]),
    # wide policies (bit-identical to the two-launch path)
    ("PDEControlGym-ReactionDiffusionPDE1D", "Neumann", "full", None, 128, 10, 50, [129, 256, 256, 1], ["relu", "relu", "tanh"]),
    ("PDEControlGym-TransportPDE1D", "Dirchilet", "collocated", None, 100, 10, 37, [1, 256, 72, 1], ["tanh", "tanh", None]),
    ("PDEControlGym-TransportPDE1D", "Neumann", "opposite", "Dirchilet", 100, 10, 16, [1, 80, 1], ["relu", None]),
])
def test_one_launch_rollout_general_cases_with_sensing_noise(kind, control, loc, stype, nx, S, B, sizes, acts):
    """Round 4: the one-launch rollout with the policy inside for Neumann actuation and scalar sensing (the policy's input is the
    one sensed value), with pre-drawn additive sensing noise (``DeviceRollout(sensing_noise=True)``: the policy reads
    obs[t] + sensing_noise[t], obs_seen records it, the plant state stays clean) and action noise -- against the policy launch +
    step launch per env-step.  Same tolerances as the Dirichlet / full-state test above; obs_seen == obs + noise exactly."""
    import pde_control_gym
    from pde_control_gym import DeviceRollout
    from pde_control_gym.src import TunedReward1D
    T, horizon = 9, 4
    dx = 1.0 / nx
    dt = 0.25 * dx * dx if "Reaction" in kind else 0.5 * dx
    nodes = nx + 1 if "Reaction" in kind else nx
    beta = np.full(nodes, 2.0, np.float32)
    net = _mlp(sizes, acts, seed=13).cuda()
    with torch.no_grad():
        net[0].weight.mul_(0.3)
    runs = {}
    for mode in (False, True):
        p = {"T": horizon * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(horizon * S, -1e3, 3e2), "normalize": True,
             "sensing_loc": loc, "control_type": control, "sensing_type": stype, "sensing_noise_func": None,
             "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 3, "control_sample_rate": S * dt,
             "batched_reset_func": lambda idx, nx_: (np.random.default_rng(5).uniform(1, 2, (len(idx), 1)).astype(np.float32)
                                                     * np.linspace(1, 1.5, nodes, dtype=np.float32)[None], np.tile(beta, (len(idx), 1)))}
        venv = pde_control_gym.make_vec(kind, num_envs=B, **p)
        venv.reset_tensor()
        venv.enable_fused_auto_reset()
        assert venv.core.can_rollout() and venv.core.obs_dim == sizes[0]
        ro = DeviceRollout(venv, FusedMLP(net), T, action_low=-1.0, action_high=1.0, action_noise=True, sensing_noise=True, one_launch=mode)
        assert ro.one_launch == mode
        g = torch.Generator().manual_seed(2)
        ro.action_noise.copy_(torch.randn(T, B, generator=g).mul(0.2).cuda())
        ro.sensing_noise.copy_(torch.randn(T + 1, B, sizes[0], generator=g).mul(0.05).cuda())
        ro.run()
        torch.cuda.synchronize()
        runs[mode] = {k: getattr(ro, k).cpu().numpy().copy() for k in ("actions", "obs", "obs_seen", "rewards", "terminated", "truncated")}
        runs[mode]["noise"] = ro.sensing_noise.cpu().numpy().copy()
        runs[mode]["time_index"] = venv.core.t["time_index"].cpu().numpy().copy()
        runs[mode]["u"] = venv.core.t["u"].cpu().numpy().copy()
        ro.run()
        torch.cuda.synchronize()
        runs[mode]["obs2"] = ro.obs.cpu().numpy().copy()
    a, b = runs[True], runs[False]
    np.testing.assert_array_equal(a["obs_seen"], a["obs"] + a["noise"])
    np.testing.assert_array_equal(b["obs_seen"], b["obs"] + b["noise"])
    if max(sizes[1:]) > 64:
        for k in ("actions", "obs", "rewards", "u", "obs2", "obs_seen"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_allclose(a["actions"][0], b["actions"][0], rtol=2e-5, atol=4e-6)
    for k in ("actions", "obs", "rewards", "u", "obs2"):
        np.testing.assert_allclose(a[k], b[k], rtol=2e-4, atol=4e-5, err_msg=k)
    for k in ("terminated", "truncated", "time_index"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(a["obs2"][0], a["obs"][T])
    assert a["terminated"].sum() > 0


@pytest.mark.gpu
def test_one_launch_rollout_falls_back_and_validates():
    from pde_control_gym import DeviceRollout
    venv = _rd_env(8, 64, 5, horizon=4)
    wide = FusedMLP(_mlp([65, 128, 1], ["tanh", None]).cuda())
    assert venv.core.policy_fits_rollout(wide) and DeviceRollout(venv, wide, 3).one_launch       # 65..256 units: cooperative evaluation
    two_out = FusedMLP(_mlp([65, 16, 2], ["tanh", None]).cuda())
    assert not venv.core.policy_fits_rollout(two_out)
    assert not DeviceRollout(venv, two_out, 3).one_launch         # two outputs: policy launch + step launch per env-step
    with pytest.raises(ValueError):
        DeviceRollout(venv, two_out, 3, one_launch=True)
    assert not DeviceRollout(venv, torch.nn.Linear(65, 1).cuda(), 3).one_launch          # a plain torch module
    # the C ABI itself refuses what the wrapper filters
    core = venv.core
    T, B, n = 3, 8, core.n
    obs = torch.zeros(T + 1, B, n, device="cuda")
    z = lambda dt: torch.zeros(T, B, dtype=dt, device="cuda")      # noqa: E731
    mism = FusedMLP(_mlp([64, 128, 1], ["tanh", None]).cuda())
    with pytest.raises(N.NativeError, match="must match the observation row"):
        core.backend.rollout1d(core.kind, core.params, core.t, obs, z(torch.float32), z(torch.float32), z(torch.uint8), z(torch.uint8), B,
                               policy=mism._net(None))
    with pytest.raises(N.NativeError, match="one command"):
        core.backend.rollout1d(core.kind, core.params, core.t, obs, z(torch.float32), z(torch.float32), z(torch.uint8), z(torch.uint8), B,
                               policy=two_out._net(None))


@pytest.mark.gpu
def test_engine_rollout_sees_in_place_weight_updates():
    """PDEBatch1D.rollout(policy=...) re-transposes weights whose parameters changed since the last call (an optimizer step
    between two rollouts), like FusedMLP.forward_into."""
    venv = _rd_env(8, 64, 5, horizon=40)
    core = venv.core
    net = _mlp([65, 16, 1], ["tanh", None], seed=2).cuda()
    pol = FusedMLP(net, clamp=(-3.0, 3.0))
    T, B, n = 2, 8, core.n
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")   # noqa: E731
    obs, act = z(T + 1, B, n), z(T, B)
    obs[0].copy_(core.t["obs"])
    first = obs[0].clone()
    core.rollout(obs, act, z(T, B), z(T, B, dt=torch.uint8), z(T, B, dt=torch.uint8), policy=pol)
    with torch.no_grad():
        want = net(first).reshape(B).clamp(-3, 3)
        np.testing.assert_allclose(act[0].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=4e-6)
        net[-1].bias.add_(0.5)                                   # "optimizer step"
        net[0].weight.mul_(1.5)
    obs[0].copy_(first)
    core.rollout(obs, act, z(T, B), z(T, B, dt=torch.uint8), z(T, B, dt=torch.uint8), policy=pol)
    with torch.no_grad():
        want2 = net(first).reshape(B).clamp(-3, 3)
    np.testing.assert_allclose(act[0].cpu().numpy(), want2.cpu().numpy(), rtol=2e-5, atol=4e-6)
    assert np.abs(want2.cpu().numpy() - want.cpu().numpy()).max() > 0.1


def test_policy_fits_rollout_limits_on_cpu_double():
    """Which policies DeviceRollout may hand to the rollout kernels (host-side rule = the C ABI's limits): layer widths, one
    output per command, LDS budget, input size, and the engine's own conditions."""
    from tests.fake_backend import FakeBackend
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    from pdecontrolgym_amd.batch_traffic import TrafficBatch

    def eng(nx, **kw):
        dx = 1.0 / nx
        return PDEBatch1D("parabolic", 100 * 0.25 * dx * dx, 0.25 * dx * dx, 1, dx, 5 * 0.25 * dx * dx, sensing_loc="full", sensing_type=None,
                          num_envs=2, device="cpu", backend=FakeBackend(), **kw)

    def pol(sizes):
        return FusedMLP(_mlp(sizes, [None] * (len(sizes) - 1)), backend=object())

    e = eng(256)
    assert e.can_rollout() and e.policy_fits_rollout(pol([257, 64, 64, 1]))
    assert e.policy_fits_rollout(pol([257, 65, 1])) and e.policy_fits_rollout(pol([257, 256, 256, 1]))     # cooperative evaluation
    assert not e.policy_fits_rollout(pol([257, 64, 2]))            # one command per instance
    assert not e.policy_fits_rollout(pol([256, 64, 1]))            # input size != row length
    assert not e.policy_fits_rollout(torch.nn.Linear(257, 1))      # not a FusedMLP
    big = eng(512)
    assert not big.policy_fits_rollout(pol([513, 64, 64, 1]))      # 131 KB of first-layer weights + 16 rows: over 160 KB
    assert big.policy_fits_rollout(pol([513, 32, 64, 1]))
    assert not eng(600).policy_fits_rollout(pol([601, 16, 1]))     # rows of more than 513 nodes
    neu = eng(64, control_type="Neumann")                          # round 4: Neumann actuation and scalar sensing roll out too
    assert neu.can_rollout() and neu.policy_fits_rollout(pol([65, 64, 1]))
    dx = 1.0 / 64
    col = PDEBatch1D("parabolic", 100 * 0.25 * dx * dx, 0.25 * dx * dx, 1, dx, 5 * 0.25 * dx * dx, sensing_loc="collocated", sensing_type=None,
                     num_envs=2, device="cpu", backend=FakeBackend())
    assert col.obs_dim == 1 and col.can_rollout() and col.policy_fits_rollout(pol([1, 64, 1])) and not col.policy_fits_rollout(pol([65, 64, 1]))
    assert not eng(64, record_history=True).can_rollout()
    assert not eng(64, state_in_obs=False).can_rollout()           # full-state sensing with a second copy of the state: two homes
    tr = TrafficBatch(240, 0.25, 500, 10, "both", 40, 0.16, 60, True, 1, num_envs=2, device="cpu", backend=FakeBackend())
    assert tr.can_rollout() and tr.policy_fits_rollout(pol([102, 64, 2])) and not tr.policy_fits_rollout(pol([102, 64, 1]))
    assert tr.policy_fits_rollout(pol([102, 256, 256, 2]))          # cooperative evaluation, two commands
    wide = TrafficBatch(240, 0.25, 1000, 10, "inlet", 40, 0.16, 60, True, 1, num_envs=2, device="cpu", backend=FakeBackend())
    assert wide.M == 101 and not wide.can_rollout()


def test_from_sb3_duck_typed_policies():
    """FusedMLP.from_sb3 picks the deterministic actor out of SB3-shaped policy objects (PPO: mlp_extractor.policy_net +
    action_net; SAC: actor.latent_pi + actor.mu + tanh) and shares their parameters."""
    import types
    from tests.fake_backend import FakeBackend
    torch.manual_seed(0)
    pi = torch.nn.Sequential(torch.nn.Linear(9, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh())
    ppo = types.SimpleNamespace(mlp_extractor=types.SimpleNamespace(policy_net=pi, value_net=torch.nn.Linear(9, 8)), action_net=torch.nn.Linear(8, 1))
    f = FusedMLP.from_sb3(ppo, clamp=(-1.0, 1.0), backend=FakeBackend())
    x = torch.randn(5, 9)
    want = ppo.action_net(pi(x)).clamp(-1, 1)
    np.testing.assert_allclose(f(x).numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert f.layers[0][0] is pi[0].weight                       # shared parameters: training updates are seen after refresh()
    lat = torch.nn.Sequential(torch.nn.Linear(9, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU())
    sac = types.SimpleNamespace(actor=types.SimpleNamespace(latent_pi=lat, mu=torch.nn.Linear(16, 2)))
    g = FusedMLP.from_sb3(sac, backend=FakeBackend())
    np.testing.assert_allclose(g(x).numpy(), torch.tanh(sac.actor.mu(lat(x))).detach().numpy(), rtol=1e-5, atol=1e-6)
    # TD3: the actor is ONE Sequential `mu` ending in Tanh, no latent_pi
    mu = torch.nn.Sequential(torch.nn.Linear(9, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1), torch.nn.Tanh())
    td3 = types.SimpleNamespace(actor=types.SimpleNamespace(mu=mu))
    h = FusedMLP.from_sb3(td3, backend=FakeBackend())
    np.testing.assert_allclose(h(x).numpy(), mu(x).detach().numpy(), rtol=1e-5, atol=1e-6)
    assert len(h.layers) == 2
    with pytest.raises(ValueError):
        FusedMLP.from_sb3(types.SimpleNamespace())
