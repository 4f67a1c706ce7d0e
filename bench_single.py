"""single_env block of bench.py: what ONE ``env.step()`` costs through the drop-in face with a batch of one.

The reference's shipped callers drive exactly this path: ``gym.make(id, **params)`` -> ``DummyVecEnv(n=1)`` -> ``PPO.learn``
(examples/transportPDE/transport1Dppo.py:59-90, examples/reactionDiffusionPDE/reactionDiffusion1Dppo.py:60-101,
examples/NavierStokes/NS2Dppo.py:29-66).  Each shape is built with ``pde_control_gym.make`` (= ``gym.make`` when gymnasium is
installed) from the reference's own parameter dictionary, stepped with float32 ``(1,)`` commands (what SB3 passes), and timed per
call with the host clock -- launch, synchronisation and result hand-over included -- beside the un-batched NumPy oracle
(oracle/single_env.py: the reference's statement sequence for one environment, trajectory kept) on the same shape in the same
process.  ``record_history`` on (the default: ``env.u`` is the whole trajectory, as in the reference) and off.
"""
from __future__ import annotations

import math
import time

import numpy as np

SHAPES = {
    # BASELINE configs[0]: TransportPDE1D nx=100, T=1 s, dt=1e-4, control every 0.1 s (1000 sub-steps per env-step)
    "transport_c1": ("PDEControlGym-TransportPDE1D", dict(T=1, dt=1e-4, X=1, dx=1e-2, control_sample_rate=0.1)),
    # examples/reactionDiffusionPDE/reactionDiffusion1Dppo.py:60-81: nx=200 (201 nodes), dt=1e-5, control every 1e-3 s (100 sub-steps)
    "parabolic_example": ("PDEControlGym-ReactionDiffusionPDE1D", dict(T=1, dt=1e-5, X=1, dx=5e-3, control_sample_rate=1e-3)),
    # one sub-step per env-step: the shape where the NumPy path is at its best (SURVEY 8d "and also S = 1")
    "transport_s1": ("PDEControlGym-TransportPDE1D", dict(T=1, dt=1e-4, X=1, dx=1e-2, control_sample_rate=1e-4)),
}


def _beta(n, amp, gamma):
    return np.array([amp * math.cos(gamma * math.acos(v)) for v in np.linspace(0, 1, n)], dtype=np.float32)


def _params(env_id, grid):
    from pde_control_gym.src import TunedReward1D
    parabolic = "Reaction" in env_id
    nx = int(round(grid["X"] / grid["dx"]))
    n = nx + (1 if parabolic else 0)
    beta = _beta(n, 50.0 if parabolic else 5.0, 8.0 if parabolic else 7.35)
    rng = np.random.default_rng(0)
    p = dict(grid, reward_class=TunedReward1D(int(round(grid["T"] / grid["dt"])), -1e3, 3e2), normalize=True, sensing_loc="full",
             control_type="Dirchilet", sensing_type=None, sensing_noise_func=lambda state: state, limit_pde_state_size=True,
             max_state_value=1e10, max_control_value=20, reset_init_condition_func=lambda nx_: np.ones(n) * rng.uniform(1, 10),
             reset_recirculation_func=lambda nx_: beta)
    return p, n, beta


def _stats(ts):
    ts = sorted(ts)
    return {"us_per_step": ts[len(ts) // 2] * 1e6, "mean_us": sum(ts) / len(ts) * 1e6, "p90_us": ts[int(len(ts) * 0.9)] * 1e6,
            "steps_timed": len(ts)}


def _time_env(step, reset, actions, seconds, max_steps):
    """Per-call host times of ``step`` (episodes restart through ``reset``, timed apart)."""
    t_reset, ts = [], []
    t0 = time.perf_counter()
    reset()
    t_reset.append(time.perf_counter() - t0)
    for k in range(8):                                   # warm-up (first launches, allocator, clocks)
        if step(actions[k % len(actions)]):
            reset()
    t_begin = time.perf_counter()
    k = 0
    while k < max_steps and time.perf_counter() - t_begin < seconds:
        a = actions[k % len(actions)]
        t0 = time.perf_counter()
        done = step(a)
        ts.append(time.perf_counter() - t0)
        k += 1
        if done:
            t0 = time.perf_counter()
            reset()
            t_reset.append(time.perf_counter() - t0)
    out = _stats(ts)
    out["reset_us"] = sorted(t_reset)[len(t_reset) // 2] * 1e6
    return out


def gpu_leg(env_id, grid, device, record_history, seconds=0.6, max_steps=3000):
    import pde_control_gym
    p, n, beta = _params(env_id, grid)
    env = pde_control_gym.make(env_id, device=str(device), record_history=record_history, **p)
    acts = np.random.default_rng(1).uniform(-1, 1, (64, 1)).astype(np.float32)

    def step(a):
        _, _, te, tr, _ = env.step(a)
        return te or tr
    out = _time_env(step, env.reset, acts, seconds, max_steps)
    out["substeps_per_env_step"] = env.unwrapped._core.substeps
    return out


def cpu_leg(env_id, grid, seconds=0.6, max_steps=3000):
    from oracle.single_env import SingleEnv1D
    p, n, beta = _params(env_id, grid)
    kind = "parabolic" if "Reaction" in env_id else "transport"
    env = SingleEnv1D(kind, grid["T"], grid["dt"], grid["X"], grid["dx"], grid["control_sample_rate"], normalize=True,
                      max_control_value=20, limit_pde_state_size=True, max_state_value=1e10,
                      reward=(int(round(grid["T"] / grid["dt"])), -1e3, 3e2))
    acts = np.random.default_rng(1).uniform(-1, 1, (64, 1)).astype(np.float32)

    def step(a):
        _, _, te, tr = env.step(a)
        return te or tr
    import warnings
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        return _time_env(step, lambda: env.reset(p["reset_init_condition_func"](0), beta), acts, seconds, max_steps)


NS_EXAMPLE = dict(T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05)          # examples/NavierStokes/NS2Dppo.py:29-50 (21 x 21, K = 2000)
NS_BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"],
         "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}


def ns_gpu_leg(device, seconds=0.6, max_steps=400):
    import pde_control_gym
    from pde_control_gym.src import NSReward
    g = NS_EXAMPLE
    nt, nx = int(round(g["T"] / g["dt"])), int(round(g["X"] / g["dx"] + 1))
    p = dict(g, action_dim=1, reward_class=NSReward(0.1), boundary_condition=NS_BC, U_ref=np.zeros((nt, nx, nx, 2)),
             action_ref=2.0 * np.ones(1000), reset_init_condition_func=lambda X: (np.zeros_like(X), np.zeros_like(X), np.zeros_like(X)))
    env = pde_control_gym.make("PDEControlGym-NavierStokes2D", device=str(device), **p)
    acts = np.random.default_rng(1).uniform(2, 4, (64, 1)).astype(np.float32)

    def step(a):
        _, _, te, tr, _ = env.step(a)
        return te or tr
    out = _time_env(step, env.reset, acts, seconds, max_steps)
    out["jacobi_sweeps_per_step"] = 2000
    return out


def ns_cpu_leg(seconds=0.6, max_steps=60):
    from oracle import pde_oracle as po
    g = NS_EXAMPLE
    nt, nx = int(round(g["T"] / g["dt"])), int(round(g["X"] / g["dx"] + 1))
    env = po.NavierStokesOracle(boundary_condition=NS_BC, U_ref=np.zeros((nt, nx, nx, 2)), action_ref=2.0 * np.ones(1000), gamma=0.1,
                                maximum_pressure_iteration=2000, **g)
    z = np.zeros((1, nx, nx))
    acts = np.random.default_rng(1).uniform(2, 4, (64, 1))

    def step(a):
        env.step(a)
        return bool(env.time_index[0] >= nt - 2)
    with np.errstate(all="ignore"):
        return _time_env(step, lambda: env.reset(z, z, z), acts, seconds, max_steps)


TRAFFIC = dict(T=240, dt=0.25, X=500, dx=10, v_steady=10, ro_steady=0.12, v_max=40, ro_max=0.16, tau=60)     # the reference notebook's freeway
TUMOR = dict(T=600, X=200, dt=1, dx=1, normalize=True, dosage_termination_threshold=0.1, t1_detection_threshold=0.8,
             t2_detection_threshold=0.16, D=0.2, rho=0.03, alpha=0.04, alpha_beta_ratio=10, k=1e5, t1_detection_radius=15,
             t1_death_radius=35, total_dosage=61.2, verbose=False)


def traffic_legs(device, seconds):
    """TrafficPDE1D (float64, 51 nodes, 'outlet' control, control_freq = 2): the drop-in face vs the (batched, B = 1) NumPy oracle."""
    import pde_control_gym
    from oracle import pde_oracle as po
    from pde_control_gym.src import TrafficARZReward
    env = pde_control_gym.make("PDEControlGym-TrafficPDE1D", device=str(device), reward_class=TrafficARZReward(), simulation_type="outlet",
                               limit_pde_state_size=True, control_freq=2, **TRAFFIC).unwrapped
    qs = 0.12 * 10
    acts = np.random.default_rng(1).uniform(0.8, 1.2, (64, 1)) * qs

    def step(a):
        _, _, d, tr, _ = env.step(a)
        return d or tr
    gpu = _time_env(step, env.reset, acts, seconds, 3000)
    orc = po.TrafficOracle(240, 0.25, 500, 10, "outlet", 40, 0.16, 60, True, 2)

    def ostep(a):
        out = orc.step(a[None])
        return bool(out[2][0] or out[3][0] or orc.time_index[0] >= 239)
    with np.errstate(all="ignore"):
        cpu = _time_env(ostep, lambda: orc.reset([0.12]), acts, seconds, 3000)
    return {"env": "PDEControlGym-TrafficPDE1D", "nx": 51, "dtype": "f64", "gpu": gpu, "numpy": cpu, "speedup": cpu["us_per_step"] / gpu["us_per_step"]}


def tumor_legs(device, seconds):
    """BrainTumor1D (float64, 201 nodes, one simulated day per step) vs the (batched, B = 1) NumPy oracle."""
    import pde_control_gym
    from oracle import pde_oracle as po
    from pde_control_gym.src import BrainTumorReward
    xs = np.linspace(0, 200, 201)
    ic = 0.8 * 1e5 * np.exp(-0.25 * (xs ** 2))
    env = pde_control_gym.make("PDEControlGym-BrainTumor1D", device=str(device), reward_class=BrainTumorReward(),
                               reset_init_condition_func=lambda X, nx: ic, **TUMOR).unwrapped
    acts = np.random.default_rng(1).uniform(0, 0.05, (64, 1))

    def step(a):
        out = env.step(float(a[0]))
        return out is None or out[2] or out[3]
    gpu = _time_env(step, env.reset, acts, seconds, 3000)
    orc = po.BrainTumorOracle(600, 1, 200, 1, 61.2)

    def ostep(a):
        out = orc.step(a)
        return bool(out[2][0] or out[3][0])
    with np.errstate(all="ignore"):
        cpu = _time_env(ostep, lambda: orc.reset(ic[None], [363.0]), acts, seconds, 3000)
    return {"env": "PDEControlGym-BrainTumor1D", "nx": 201, "dtype": "f64", "gpu": gpu, "numpy": cpu, "speedup": cpu["us_per_step"] / gpu["us_per_step"]}


def single_env_block(device, seconds=0.6):
    """{shape: {"gpu": {...}, "gpu_no_history": {...}, "numpy": {...}, "speedup": gpu-vs-numpy}} + the crossover sub-step count."""
    out = {}
    for name, (env_id, grid) in SHAPES.items():
        try:
            e = {"env": env_id, "nx": int(round(grid["X"] / grid["dx"])),
                 "gpu": gpu_leg(env_id, grid, device, True, seconds), "gpu_no_history": gpu_leg(env_id, grid, device, False, seconds),
                 "numpy": cpu_leg(env_id, grid, seconds)}
            e["speedup"] = e["numpy"]["us_per_step"] / e["gpu"]["us_per_step"]
            out[name] = e
        except Exception as ex:          # keep the headline line alive
            out[name] = {"error": repr(ex)}
    try:
        e = {"env": "PDEControlGym-NavierStokes2D", "nx": 21, "dtype": "f64", "gpu": ns_gpu_leg(device, seconds), "numpy": ns_cpu_leg(seconds)}
        e["speedup"] = e["numpy"]["us_per_step"] / e["gpu"]["us_per_step"]
        out["ns2d_example"] = e
    except Exception as ex:
        out["ns2d_example"] = {"error": repr(ex)}
    for name, legs in (("traffic_example", traffic_legs), ("tumor_example", tumor_legs)):
        try:
            out[name] = legs(device, seconds)
        except Exception as ex:
            out[name] = {"error": repr(ex)}
    # crossover: per-sub-step cost of both paths from the S = 1 and S = 1000 transport shapes (cost = a + b S)
    try:
        def line(leg):
            c1, c1000 = out["transport_s1"][leg]["us_per_step"], out["transport_c1"][leg]["us_per_step"]
            b = (c1000 - c1) / 999.0
            return c1 - b, b
        (ag, bg), (an, bn) = line("gpu"), line("numpy")
        out["crossover"] = {"gpu_us": {"fixed": ag, "per_substep": bg}, "numpy_us": {"fixed": an, "per_substep": bn},
                            "substeps_above_which_gpu_wins": (max(0.0, (ag - an) / (bn - bg)) if bn > bg else None),
                            "note": "TransportPDE1D nx=100, one environment: cost per env.step() = fixed + per_substep * S"}
    except Exception as ex:
        out["crossover"] = {"error": repr(ex)}
    return out


if __name__ == "__main__":
    import json
    import sys
    import torch
    dev = torch.device("cuda", 0)
    print(json.dumps(single_env_block(dev, float(sys.argv[1]) if len(sys.argv) > 1 else 0.6), indent=1))
