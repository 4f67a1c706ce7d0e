#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the REFERENCE's own
NumPy code (lukebhan/PDEControlGym checkout at /root/reference) in the build container.

Only inputs and expected outputs are stored (small .npz files); no reference source, bytecode or
pickles.  The reference cannot be imported as shipped (SyntaxError in pde_control_gym/__init__.py,
and gymnasium is not installed) so a metadata-only shim is used: it supplies ``gymnasium.Env``,
``spaces.Box`` and ``register`` as no-ops and mounts the reference's package directory under a
synthetic parent so the broken ``__init__`` is skipped.  The shim contains no arithmetic: every
number written below is produced by the reference's own step()/reset()/reward() code.

Run:  python tests/golden/make_golden.py      (needs /root/reference; NOT run on the GPU box)
"""
import importlib
import math
import os
import sys
import types
import warnings

import numpy as np

REF = os.environ.get("PDEGYM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:          # `python tests/golden/make_golden.py` from anywhere: tests.cases must be importable
    sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore", category=DeprecationWarning)


def import_reference():
    import matplotlib
    matplotlib.use("Agg")
    gym = types.ModuleType("gymnasium")

    class Env:
        def __init__(self):
            pass

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape) if shape is not None else np.asarray(low, dtype=dtype)
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape) if shape is not None else np.asarray(high, dtype=dtype)
            self.shape, self.dtype = self.low.shape, np.dtype(dtype)

    gym.Env, gym.Wrapper = Env, Wrapper
    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    gym.spaces = spaces
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **k: None
    envs.registration = reg
    gym.envs = envs
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces,
                        "gymnasium.envs": envs, "gymnasium.envs.registration": reg})
    pkg = types.ModuleType("pde_control_gym")
    pkg.__path__ = [os.path.join(REF, "pde_control_gym")]
    sys.modules["pde_control_gym"] = pkg
    sys.dont_write_bytecode = True
    return importlib.import_module("pde_control_gym.src")


def cheb_beta(x, gamma, amp):
    beta = np.zeros(len(x), dtype=np.float32)
    for idx, val in enumerate(x):
        beta[idx] = amp * math.cos(gamma * math.acos(val))
    return beta


def _as_control(a, action_as):
    """The Python object handed to the reference's step(): what decides NumPy's promotion of control_update/normalize."""
    if action_as == "f32arr":
        return np.array([a], dtype=np.float32)       # what SB3 passes
    if action_as == "npf64":
        return np.float64(a)                         # e.g. a backstepping controller's dot product
    if action_as == "pyfloat":
        return float(a)
    if action_as == "pyint":
        return int(a)
    raise ValueError(action_as)


def run_1d(src, cls_name, kw, init, beta, actions, reward_args, extra=None, action_as="f32arr"):
    """Drive one reference 1D env (by default with float32 (1,) actions like SB3 does); record everything."""
    cls = getattr(src, cls_name)
    kw = dict(kw)
    kw["reward_class"] = src.TunedReward1D(*reward_args)
    kw["sensing_noise_func"] = lambda s: s
    kw["reset_init_condition_func"] = lambda nx: init
    kw["reset_recirculation_func"] = lambda nx: beta
    env = cls(**kw)
    obs0, _ = env.reset()
    obs, rew, term, trunc, tidx, rows = [np.array(obs0, dtype=np.float32)], [], [], [], [], []
    for a in actions:
        o, r, te, tr, _ = env.step(_as_control(a, action_as))
        obs.append(np.array(o, dtype=np.float32).reshape(-1))
        rew.append(np.float64(r))
        term.append(te)
        trunc.append(tr)
        tidx.append(env.time_index)
        rows.append(np.array(env.u[env.time_index], dtype=np.float32))
    obs[0] = obs[0].reshape(-1)
    return dict(obs=np.stack(obs), reward=np.array(rew), terminate=np.array(term), truncate=np.array(trunc),
                time_index=np.array(tidx), rows=np.stack(rows), init=np.asarray(init), beta=np.asarray(beta),
                actions=np.asarray(actions, dtype=np.float32 if action_as == "f32arr" else np.float64),
                reward_args=np.array(reward_args, dtype=np.float64), action_as=np.array(action_as))


def pack(prefix, d, store):
    for k, v in d.items():
        store[f"{prefix}/{k}"] = v


def gen_transport(src, out=HERE):
    store = {}
    rng = np.random.default_rng(1234)
    # F-H1: BASELINE config 1 (nx=100, T=1, dt=1e-4, S=1000)
    base = dict(T=1, dt=1e-4, X=1, dx=1e-2, normalize=False, sensing_loc="full", control_type="Dirchilet",
                sensing_type=None, limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
                control_sample_rate=0.1)
    nx = 100
    beta = cheb_beta(np.linspace(0, 1, nx), 7.35, 5)
    acts = rng.uniform(-1, 1, 10).astype(np.float32)
    pack("H1", run_1d(src, "TransportPDE1D", base, np.ones(nx) * 5.0, beta, acts, (10000, -1e3, 3e2)), store)
    # F-H2: Neumann + normalize, every sensing mode
    for name, (ct, sl, st) in {
        "neu_full": ("Neumann", "full", None), "neu_col": ("Neumann", "collocated", None),
        "neu_opp_neu": ("Neumann", "opposite", "Neumann"), "neu_opp_dir": ("Neumann", "opposite", "Dirchilet"),
        "dir_col": ("Dirchilet", "collocated", None), "dir_opp_neu": ("Dirchilet", "opposite", "Neumann"),
        "dir_opp_dir": ("Dirchilet", "opposite", "Dirchilet"),
    }.items():
        kw = dict(base, T=0.3, control_type=ct, sensing_loc=sl, sensing_type=st, normalize=True, control_sample_rate=0.05)
        acts = rng.uniform(-1, 1, 6).astype(np.float32)
        pack(f"H2_{name}", run_1d(src, "TransportPDE1D", kw, np.ones(nx) * 3.0, beta, acts, (3000, -1e3, 3e2)), store)
    # F-H3: BASELINE config 3 shape (nx=512, dt=0.5dx, S=100), smooth random IC / beta
    nx = 512
    dx = 1.0 / 512
    dt = 0.5 * dx
    x = np.linspace(0, 1, nx)
    init = (2.0 + np.sin(2 * np.pi * x * rng.uniform(0.5, 2)) * rng.uniform(0.5, 3)).astype(np.float32)
    beta = cheb_beta(x, rng.uniform(7, 7.7), 5)
    kw = dict(base, T=700 * dt, dt=dt, dx=dx, control_sample_rate=100 * dt)
    acts = rng.uniform(-1, 1, 8).astype(np.float32)     # 7 steps reach nt-1, the 8th is a post-terminal call
    pack("H3", run_1d(src, "TransportPDE1D", kw, init, beta, acts, (700, -1e3, 3e2)), store)
    # F-R: reward edge cases. (a) S=30 (<100: look-back wraps into zero rows, then hits non-step-end rows),
    # clipped last step, terminate with ||u|| >= 20; (b) truncation through a small max_state_value
    nx = 100
    beta = cheb_beta(np.linspace(0, 1, nx), 7.35, 5)
    kw = dict(base, T=0.0400, dt=1e-4, control_sample_rate=30e-4)     # nt=401, S=30 -> 14 steps, last has 10 sub-steps
    acts = rng.uniform(-1, 1, 15).astype(np.float32)
    pack("R_s30", run_1d(src, "TransportPDE1D", kw, np.ones(nx) * 8.0, beta, acts, (400, -1e3, 3e2)), store)
    kw = dict(base, T=0.0400, dt=1e-4, control_sample_rate=30e-4)
    pack("R_s30_small", run_1d(src, "TransportPDE1D", kw, np.ones(nx) * 0.5, beta, acts, (400, -1e3, 3e2)), store)
    kw = dict(base, T=1, max_state_value=46.5)     # ||u|| = 46.08, 46.96 -> truncates at step 2 (and 46.38 < 46.5 at step 3)
    acts = (np.ones(10) * 0.9).astype(np.float32)
    pack("R_trunc", run_1d(src, "TransportPDE1D", kw, np.ones(nx) * 5.0, beta, acts, (10000, -1e3, 3e2)), store)
    # tiny horizon: nt <= 100, negative look-back index wraps onto rows that HAVE been written
    # (the reference raises IndexError when t-100 < -nt, so S must be >= 100-nt+1)
    kw = dict(base, T=0.0090, dt=1e-4, control_sample_rate=15e-4)      # nt=91, S=15: t-100 -> rows 6,21,36,...
    acts = rng.uniform(-1, 1, 6).astype(np.float32)
    pack("R_tiny", run_1d(src, "TransportPDE1D", kw, np.ones(nx) * 2.0, beta, acts, (90, -1e3, 3e2)), store)
    np.savez_compressed(os.path.join(out, "transport.npz"), **store)


def gen_parabolic(src, out=HERE):
    store = {}
    rng = np.random.default_rng(4321)
    base = dict(T=1, dt=1e-5, X=1, dx=5e-3, normalize=False, sensing_loc="full", control_type="Dirchilet",
                sensing_type=None, limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
                control_sample_rate=1e-3)
    # F-P1: shipped example config (nx=200, S=100), 20 steps
    nx = 200
    beta = cheb_beta(np.linspace(0, 1, nx + 1), 8, 50)
    acts = rng.uniform(-1, 1, 20).astype(np.float32)
    pack("P1", run_1d(src, "ReactionDiffusionPDE1D", base, np.ones(nx + 1) * 4.0, beta, acts, (100000, -1e3, 3e2)), store)
    # F-P2: BASELINE config 2 shape (nx=256, F=0.25), S in {1, 100}; Dirichlet and Neumann+normalize
    nx = 256
    dx = 1.0 / 256
    dt = 0.25 * dx * dx
    x = np.linspace(0, 1, nx + 1)
    beta = cheb_beta(x, 8.2, 50)
    init = (np.ones(nx + 1) * 6.5).astype(np.float32)
    for name, S, ct, norm, nsteps, T in [("P2_s100", 100, "Dirchilet", False, 12, 1000 * dt),
                                         ("P2_s1", 1, "Dirchilet", False, 12, 1000 * dt),
                                         ("P2_s100_neu", 100, "Neumann", False, 8, 1000 * dt),
                                         ("P2_s1_neu", 1, "Neumann", False, 150, 1000 * dt),
                                         # normalize also scales the Neumann neighbour (x20 per sub-step): keep it short
                                         ("P2_s1_neu_norm", 1, "Neumann", True, 6, 1000 * dt),
                                         ("P2_s100_dir_norm", 100, "Dirchilet", True, 6, 1000 * dt)]:
        kw = dict(base, T=T, dt=dt, dx=dx, control_sample_rate=S * dt, control_type=ct, normalize=norm)
        acts = rng.uniform(-1, 1, nsteps).astype(np.float32)
        pack(name, run_1d(src, "ReactionDiffusionPDE1D", kw, init, beta, acts, (1000, -1e3, 3e2)), store)
    # sensing variants
    for name, (ct, sl, st) in {"col_neu": ("Neumann", "collocated", None), "col_dir": ("Dirchilet", "collocated", None),
                               "opp_neu": ("Dirchilet", "opposite", "Neumann")}.items():
        kw = dict(base, T=600 * dt, dt=dt, dx=dx, control_sample_rate=50 * dt, control_type=ct, sensing_loc=sl, sensing_type=st)
        acts = rng.uniform(-1, 1, 5).astype(np.float32)
        pack(f"P3_{name}", run_1d(src, "ReactionDiffusionPDE1D", kw, init, beta, acts, (600, -1e3, 3e2)), store)
    np.savez_compressed(os.path.join(out, "parabolic.npz"), **store)


def gen_kat(src, out=HERE):
    """Published known answers (backstepping episodes, notebook stored outputs)."""
    store = {}
    # transport: examples/transportPDE/transport1Dbackstepping.py:22-36,48-99
    T, dt, dx, X = 5, 1e-4, 1e-2, 1
    nx = 100

    def kernel_transport(theta):
        kappa = np.zeros(len(theta))
        for i in range(len(theta)):
            s = 0
            for j in range(i):
                s += (kappa[i - j] * theta[j]) * dx
            kappa[i] = s - theta[i]
        return np.flip(kappa)

    beta = cheb_beta(np.linspace(0, 1, nx), 7.35, 5)
    kern = kernel_transport(cheb_beta(np.linspace(dx, X, nx), 7.35, 5))
    for u0 in (1, 10):
        kw = dict(T=T, dt=dt, X=X, dx=dx, reward_class=src.TunedReward1D(int(round(T / dt)), -1e3, 3e2), normalize=False,
                  sensing_loc="full", control_type="Dirchilet", sensing_type=None, sensing_noise_func=lambda s: s,
                  limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
                  reset_init_condition_func=lambda n, u0=u0: np.ones(n) * u0, reset_recirculation_func=lambda n: beta,
                  control_sample_rate=0.1)
        env = src.TransportPDE1D(**kw)
        obs, _ = env.reset()
        te = tr = False
        total, l2, acts, rews = 0.0, 0.0, [], []
        while not te and not tr:
            a = 0
            for i in range(len(obs)):
                a += kern[i] * obs[i]
            a = a * 1e-2
            obs, r, te, tr, _ = env.step(a)
            total += r
            l2 += np.linalg.norm(obs)
            acts.append(a)
            rews.append(r)
        pack(f"T_u{u0}", dict(total=np.float64(total), sum_l2=np.float64(l2), actions=np.array(acts, dtype=np.float64),
                              rewards=np.array(rews, dtype=np.float64), kernel=kern, beta=beta, last_obs=np.array(obs)), store)
        print("KAT transport", u0, total, l2)

    # parabolic: examples/reactionDiffusionPDE/reactionDiffusion1DBackstepping.py:22-39,51-102
    T, dt, dx, X = 1, 1e-5, 5e-3, 1
    nx = 200

    def kernel_parabolic(a):
        k = np.zeros((len(a), len(a)))
        k[1][1] = -(a[1] + a[0]) * dx / 4
        for i in range(1, len(a) - 1):
            k[i + 1][0] = 0
            k[i + 1][i + 1] = k[i][i] - dx / 4.0 * (a[i - 1] + a[i])
            k[i + 1][i] = k[i][i] - dx / 2 * a[i]
            for j in range(1, i):
                k[i + 1][j] = -k[i - 1][j] + k[i][j + 1] + k[i][j - 1] + a[j] * (dx ** 2) * (k[i][j + 1] + k[i][j - 1]) / 2
        return k

    beta = cheb_beta(np.linspace(0, 1, nx + 1), 8, 50)
    kern = kernel_parabolic(cheb_beta(np.linspace(dx, X, nx), 8, 50))   # notebook cell 11 grid
    for u0 in (1, 10):
        kw = dict(T=T, dt=dt, X=X, dx=dx, reward_class=src.TunedReward1D(int(round(T / dt)), -1e3, 3e2), normalize=False,
                  sensing_loc="full", control_type="Dirchilet", sensing_type=None, sensing_noise_func=lambda s: s,
                  limit_pde_state_size=True, max_state_value=1e10, max_control_value=20,
                  reset_init_condition_func=lambda n, u0=u0: np.ones(n + 1) * u0, reset_recirculation_func=lambda n: beta,
                  control_sample_rate=0.001)
        env = src.ReactionDiffusionPDE1D(**kw)
        obs, _ = env.reset()
        te = tr = False
        total, l2, acts, rews = 0.0, 0.0, [], []
        krow = kern[-1]
        while not te and not tr:
            m = min(len(krow), len(obs) - 1)
            a = sum(krow[0:m] * obs[0:m]) * dx
            obs, r, te, tr, _ = env.step(a)
            total += r
            l2 += np.linalg.norm(obs)
            acts.append(a)
            rews.append(r)
        pack(f"P_u{u0}", dict(total=np.float64(total), sum_l2=np.float64(l2), actions=np.array(acts, dtype=np.float64),
                              rewards=np.array(rews, dtype=np.float64), kernel_row=krow, beta=beta, last_obs=np.array(obs)), store)
        print("KAT parabolic", u0, total, l2)
    np.savez_compressed(os.path.join(out, "kat.npz"), **store)


def gen_mixed(src, out=HERE):
    """float64 plant parameter and/or float64 / Python-scalar control inputs: NumPy then evaluates parts of the update in
    double and rounds once when the row is stored (hyperbolic.py:146-155, parabolic.py:143-150).  Case table: tests/cases.py
    MIXED_CASES (kwargs) -- here only the data."""
    from tests.cases import MIXED_CASES
    store = {}
    rng = np.random.default_rng(8642)
    for name, (kind, kw, action_as, beta_kind, nsteps) in MIXED_CASES.items():
        nx = int(round(kw["X"] / kw["dx"]))
        n = nx + (1 if kind == "parabolic" else 0)
        x = np.linspace(0, 1, n)
        if beta_kind == "ones64":            # docs/source/guide/quickstart.rst:27-28
            beta = np.ones(n)
        elif beta_kind == "cos64":           # an un-cast NumPy expression is float64
            beta = (50 if kind == "parabolic" else 5) * np.cos(rng.uniform(7, 8.5) * np.arccos(x))
        elif beta_kind == "int":
            beta = np.arange(n) % 3
        else:                                # "cos32": the examples' float32 beta (only the control input is float64)
            beta = cheb_beta(x, 7.35 if kind == "transport" else 8.0, 5 if kind == "transport" else 50)
        init = np.ones(n) * rng.uniform(1, 10) if name.startswith("Q_quick") else \
            (rng.uniform(1, 10) * (1 + 0.3 * np.sin(2 * np.pi * x * rng.uniform(0.5, 3))))
        if action_as == "pyint":
            acts = np.zeros(nsteps)          # quickstart.rst:68: env.step(0)
        else:
            acts = rng.uniform(-1, 1, nsteps)
        nt1 = int(round(kw["T"] / kw["dt"]))
        cls = "ReactionDiffusionPDE1D" if kind == "parabolic" else "TransportPDE1D"
        with np.errstate(all="ignore"):
            pack(name, run_1d(src, cls, kw, init, beta, acts, (nt1, -1e-4, 1e2) if name.startswith("Q_quick") else (nt1, -1e3, 3e2),
                              action_as=action_as), store)
    np.savez_compressed(os.path.join(out, "mixed.npz"), **store)


NS_BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"],
         "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}


def gen_ns(src, out=HERE):
    store = {}
    # F-N1: the reference's own golden trajectory examples/NavierStokes/target.npz (NS2Dppo.py:21-50)
    tgt = np.load(os.path.join(REF, "examples/NavierStokes/target.npz"))
    ut, vt = tgt["u"], tgt["v"]
    Uref = np.stack([ut, vt], axis=-1)
    kw = dict(T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05, action_dim=1, reward_class=src.NSReward(0.1), normalize=False,
              reset_init_condition_func=lambda X: (ut[0].copy(), vt[0].copy(), np.zeros_like(ut[0])),
              boundary_condition=NS_BC, U_ref=Uref, action_ref=2.0 * np.ones(1000))
    env = src.NavierStokes2D(**kw)
    env.reset()
    rews, frames = [], {}
    keep = [0, 1, 2, 50, 120, 199]
    acts = ut[1:200, -1, 10].copy()          # = 4 - 0.01 t up to rounding; the stored values are the exact inputs
    for t in range(1, 200):
        obs, r, te, tr, _ = env.step(acts[t - 1])
        rews.append(r)
    U = env.U
    assert np.array_equal(U[:200, :, :, 0], ut) and np.array_equal(U[:200, :, :, 1], vt), "reference no longer reproduces target.npz"
    for t in keep:
        frames[f"u{t}"] = ut[t]
        frames[f"v{t}"] = vt[t]
    pack("N1", dict(rewards=np.array(rews), p_final=env.p, keep=np.array(keep), actions=acts, **frames), store)
    # a different reference trajectory so the reward is non-trivial
    env2 = src.NavierStokes2D(**dict(kw, U_ref=0.5 * Uref))
    env2.reset()
    rews2 = [env2.step(acts[t - 1])[1] for t in range(1, 6)]
    pack("N1b", dict(rewards=np.array(rews2)), store)

    # F-N2: mixed boundary conditions, random smooth IC, K=50
    rng = np.random.default_rng(99)
    for name, n, bc in [("N2_32", 32, {"upper": ["Controllable", "Neumann"], "lower": ["Dirchilet", "Controllable"],
                                          "left": ["Neumann", "Dirchilet"], "right": ["Dirchilet", "Neumann"]}),
                        ("N2_64", 64, {"upper": ["Neumann", "Neumann"], "lower": ["Controllable", "Dirchilet"],
                                          "left": ["Controllable", "Controllable"], "right": ["Neumann", "Dirchilet"]}),
                        ("N2_48", 48, NS_BC)]:
        dx = 1.0 / (n - 1)
        dt = 0.2 * 0.5 * dx * dx / 0.1
        xs = np.linspace(0, 1, n)
        Xg, Yg = np.meshgrid(xs, xs)
        u0 = np.sin(2 * np.pi * Xg) * np.cos(np.pi * Yg) * rng.uniform(0.5, 2) + rng.uniform(-1, 1)
        v0 = np.cos(np.pi * Xg) * np.sin(2 * np.pi * Yg) * rng.uniform(0.5, 2) + rng.uniform(-1, 1)
        p0 = rng.uniform(-1, 1, (n, n))
        nt = 10
        Uref = rng.uniform(-1, 1, (nt, n, n, 2))
        aref = rng.uniform(1, 3, nt)
        kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, action_dim=1, reward_class=src.NSReward(0.1), normalize=False,
                  reset_init_condition_func=lambda X: (u0.copy(), v0.copy(), p0.copy()), boundary_condition=bc,
                  U_ref=Uref, action_ref=aref, maximum_pressure_iteration=50)
        env = src.NavierStokes2D(**kw)
        assert env.nx == n
        env.reset()
        acts = rng.uniform(2, 4, 3)
        obs_l, p_l, r_l = [], [], []
        for a in acts:
            obs, r, te, tr, _ = env.step(a)
            obs_l.append(np.array(obs))
            p_l.append(np.array(env.p))
            r_l.append(r)
        pack(name, dict(u0=u0, v0=v0, p0=p0, U_ref=Uref, action_ref=aref, actions=acts, obs=np.stack(obs_l), p=np.stack(p_l),
                        rewards=np.array(r_l), dx=np.float64(dx), dt=np.float64(dt), nt=np.int64(nt),
                        bc=np.array([bc[k][i] for k in ("upper", "lower", "left", "right") for i in (0, 1)])), store)

    # F-N3: BASELINE config 4 shape (128x128, K=50): checksums + sampled points only
    n = 128
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    cu, cv, cp = 1.7, -2.3, 0.4
    nt = 10
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, action_dim=1, reward_class=src.NSReward(0.1), normalize=False,
              reset_init_condition_func=lambda X: (cu * np.ones_like(X), cv * np.ones_like(X), cp * np.ones_like(X)),
              boundary_condition=NS_BC, U_ref=np.zeros((nt, n, n, 2)), action_ref=2.0 * np.ones(nt), maximum_pressure_iteration=50)
    env = src.NavierStokes2D(**kw)
    env.reset()
    pts = rng.integers(0, n, (16, 2))
    acts = np.array([3.1, 2.4])
    sums, samples, rews = [], [], []
    for a in acts:
        obs, r, te, tr, _ = env.step(a)
        sums.append([np.linalg.norm(obs[..., 0]), np.linalg.norm(obs[..., 1]), np.linalg.norm(env.p),
                     obs.min(), obs.max(), env.p.min(), env.p.max()])
        samples.append(np.stack([obs[pts[:, 0], pts[:, 1], 0], obs[pts[:, 0], pts[:, 1], 1], env.p[pts[:, 0], pts[:, 1]]], axis=-1))
        rews.append(r)
    pack("N3", dict(ic=np.array([cu, cv, cp]), actions=acts, sums=np.array(sums), pts=pts, samples=np.stack(samples),
                    rewards=np.array(rews), dx=np.float64(dx), dt=np.float64(dt), nt=np.int64(nt)), store)
    np.savez_compressed(os.path.join(out, "ns2d.npz"), **store)


def gen_traffic(src, out=HERE):
    """TrafficPDE1D (environments1d/traffic_arz_env.py) with TrafficARZReward: the shipped notebook configuration
    (examples/TrafficPDE1D/*.ipynb cell 3: T=240, dt=0.25, dx=10, X=500, tau=60, v_max=40, ro_max=0.16)."""
    import contextlib
    import io
    import random
    store = {}
    rng = np.random.default_rng(77)
    for name, sim, cf, nact, nsteps, limit in [("inlet", "inlet", 1, 1, 60, True), ("outlet", "outlet", 1, 1, 60, True),
                                               ("both", "both", 1, 2, 60, True), ("train", "outlet-train", 2, 1, 60, True),
                                               ("outlet_cf3", "outlet", 3, 1, 40, False), ("long", "inlet", 1, 1, 1000, True)]:
        kw = dict(T=240, dt=0.25, X=500, dx=10, reward_class=src.TrafficARZReward(), simulation_type=sim, v_steady=10,
                  ro_steady=0.12, v_max=40, ro_max=0.16, tau=60, limit_pde_state_size=limit, control_freq=cf)
        random.seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            env = src.TrafficPDE1D(**kw)
            qs_clip = env.qs
            random.seed(13)
            obs0, _ = env.reset()
        acts = rng.uniform(0.7, 1.3, (nsteps, nact)) * env.qs
        obs, rew, done, trunc, tim = [np.array(obs0)], [], [], [], []
        for a in acts:
            with contextlib.redirect_stdout(io.StringIO()):
                o, r, d, t, _ = env.step(a)
            obs.append(np.array(o))
            rew.append(r)
            done.append(bool(d))
            trunc.append(bool(t))
            tim.append(env.time_index)
        obs = np.stack(obs)
        if name == "long":      # keep the file small: every 50th observation
            keep = np.arange(0, nsteps + 1, 50)
            obs = obs[keep]
            store[f"{name}/keep"] = keep
        pack(name, dict(obs=obs, reward=np.array(rew), done=np.array(done), trunc=np.array(trunc), time=np.array(tim),
                        actions=acts, rs=np.float64(env.rs), qs_clip=np.float64(qs_clip), control_freq=np.int64(cf),
                        limit=np.bool_(limit), sim=np.array(sim)), store)
    np.savez_compressed(os.path.join(out, "traffic.npz"), **store)


def tumor_ic(X, nx):
    """examples/BrainTumor1D notebook: 0.8 k exp(-0.25 x^2)."""
    xs = np.linspace(0, X, nx)
    return 0.8 * 1e5 * np.exp(-0.25 * (xs ** 2))


TUMOR_KW = dict(X=200, dt=1, dx=1, normalize=True, dosage_termination_threshold=0.1, t1_detection_threshold=0.8,
                t2_detection_threshold=0.16, D=0.2, rho=0.03, alpha=0.04, alpha_beta_ratio=10, k=1e5,
                t1_detection_radius=15, t1_death_radius=35, total_dosage=61.2, verbose=False)


def gen_tumor(src, out=HERE):
    """BrainTumor1D + BrainTumorReward + TherapyWrapper (environments1d/brain_tumor_env.py, rewards/brain_tumor_reward.py)
    on the shipped notebook configuration (examples/BrainTumor1D: T=600, X=200, dt=dx=1, total_dosage=61.2)."""
    import importlib
    bt = importlib.import_module("pde_control_gym.src.environments1d.brain_tumor_env")
    br = importlib.import_module("pde_control_gym.src.rewards.brain_tumor_reward")
    store = {}
    rng = np.random.default_rng(2024)
    # ---- raw environment episodes: (name, T, t_benchmark, dose range)
    for name, T, tb, hi in [("raw", 600, 300, 0.08), ("toxic", 600, 250, 0.3), ("nobench", 600, None, 0.1),
                            ("term_therapy", 230, 200, 0.004), ("term_post", 300, 200, 0.25)]:
        env = bt.BrainTumor1D(T=T, reward_class=br.BrainTumorReward(), reset_init_condition_func=tumor_ic, **TUMOR_KW)
        env.t_benchmark = tb
        obs0, _ = env.reset()
        acts, rew, term, trunc, stage = [], [], [], [], []
        while True:
            a = float(rng.uniform(0, hi))
            o, r, te, tr, info = env.step(a)
            acts.append(a)
            rew.append(float(r))
            term.append(bool(te))
            trunc.append(bool(tr))
            stage.append({"Growth": 0, "Therapy": 1, "Post-Therapy": 2}[info["stage"]])
            if te or tr:
                break
        n = len(acts)
        keep = np.unique(np.concatenate([np.arange(0, n + 1, 16), [n]]))
        pack(name, dict(T=np.int64(T), t_benchmark=np.float64(np.nan if tb is None else tb), actions=np.array(acts),
                        reward=np.array(rew), term=np.array(term), trunc=np.array(trunc), stage=np.array(stage),
                        keep=keep, rows=env.u[keep].copy(), t1_idx=env.t1_radius_idx_vs_time[: n + 1].copy(),
                        dosage=env.dosage_vs_time[: n + 1].copy(),
                        days=np.array([env.growthDays, env.therapyDays, env.postTherapyDays, env.simulationDays,
                                       -1 if env.cDeathDay is None else env.cDeathDay]),
                        first=np.array([-1 if env.firstTherapyDay is None else env.firstTherapyDay,
                                        -1 if env.firstPostTherapyDay is None else env.firstPostTherapyDay]),
                        remaining=np.float64(env.remaining_dosage)), store)
    # ---- wrapper flows: benchmark(), reset() through the growth stage, constant daily fraction
    for name, weekends, frac in [("wrap_week", True, 2.0 / 61.2), ("wrap_daily", False, 1.8 / 61.2), ("wrap_hypo", True, 0.2)]:
        env = bt.BrainTumor1D(T=600, reward_class=br.BrainTumorReward(), reset_init_condition_func=tumor_ic, **TUMOR_KW)
        w = bt.TherapyWrapper(env, weekends=weekends, verbose=False)
        tb = w.benchmark()
        obs, _ = w.reset()
        rows, rew, term, trunc, tidx = [np.array(obs)], [], [], [], [env.time_index]
        while True:
            o, r, te, tr, info = w.step(frac)
            rows.append(np.array(o))
            rew.append(float(r))
            term.append(bool(te))
            trunc.append(bool(tr))
            tidx.append(env.time_index)
            if te or tr:
                break
        pack(name, dict(weekends=np.bool_(weekends), frac=np.float64(frac), t_benchmark=np.int64(tb), rows=np.stack(rows),
                        reward=np.array(rew), term=np.array(term), trunc=np.array(trunc), time_index=np.array(tidx),
                        days=np.array([env.growthDays, env.therapyDays, env.postTherapyDays, env.simulationDays,
                                       -1 if env.cDeathDay is None else env.cDeathDay]),
                        calls=np.int64(w.treatment_calls), violations=np.int64(w.soft_constraint_violations),
                        dosage=env.dosage_vs_time.copy(), t1_idx=env.t1_radius_idx_vs_time.copy()), store)
    np.savez_compressed(os.path.join(out, "tumor.npz"), **store)


FILES = {"transport": "transport.npz", "parabolic": "parabolic.npz", "kat": "kat.npz", "mixed": "mixed.npz", "ns": "ns2d.npz",
         "traffic": "traffic.npz", "tumor": "tumor.npz"}


GEN = {"transport": gen_transport, "parabolic": gen_parabolic, "kat": gen_kat, "mixed": gen_mixed, "ns": gen_ns,
       "traffic": gen_traffic, "tumor": gen_tumor}


def check(which=None):
    """Regenerate into a scratch directory and compare with the committed fixtures: same keys, same dtypes, same shapes, same
    BITS.  Returns the list of differences (empty = the committed files are exactly what the generator writes today)."""
    import tempfile
    committed = HERE
    diffs = []
    with tempfile.TemporaryDirectory() as tmp:
        src = import_reference()
        for w in (which or list(GEN)):
            GEN[w](src, tmp)
            a, b = np.load(os.path.join(committed, FILES[w]), allow_pickle=False), np.load(os.path.join(tmp, FILES[w]), allow_pickle=False)
            for k in sorted(set(a.files) | set(b.files)):
                if k not in a.files or k not in b.files:
                    diffs.append(f"{FILES[w]}: key {k} only in the {'generator output' if k in b.files else 'committed file'}")
                elif a[k].dtype != b[k].dtype or a[k].shape != b[k].shape or a[k].tobytes() != b[k].tobytes():
                    diffs.append(f"{FILES[w]}: {k} differs")
    return diffs


if __name__ == "__main__":
    if "--check" in sys.argv[1:]:
        d = check([w for w in sys.argv[1:] if w != "--check"] or list(GEN))
        print("\n".join(d) if d else "fixtures == generator output")
        sys.exit(1 if d else 0)
    src = import_reference()
    which = sys.argv[1:] or ["transport", "parabolic", "kat", "mixed", "ns", "traffic", "tumor"]
    store_meta = dict(numpy=np.__version__)
    for w in which:
        GEN[w](src)
        print("wrote", w)
    with open(os.path.join(out, "VERSIONS.txt"), "w") as f:
        f.write(f"numpy {np.__version__}\nreference snapshot 2026-01-09 (lukebhan/PDEControlGym)\n")
