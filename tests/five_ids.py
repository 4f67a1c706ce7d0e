"""Parameter dictionaries of the reference's five registered environment ids (small horizons), shared by tests/test_real_sb3.py (the
real gymnasium / SB3, skipped in this image) and tests/test_host_api.py (the same dictionaries through ``pde_control_gym.make`` on the
CPU double and on HIP: they must at least build, reset and step wherever the real packages are absent)."""
import numpy as np

from pde_control_gym.src import NSReward, TunedReward1D

NX = 100
BETA = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, NX)))).astype(np.float32)
ICS = [np.linspace(1.0, 2.0 + k, NX).astype(np.float32) for k in range(8)]


def _backend(kind):
    if kind == "double":
        from tests.fake_backend import FakeBackend
        return dict(device="cpu", backend=FakeBackend())
    import torch
    assert torch.cuda.is_available(), "hip run without a GPU"
    return dict(device="cuda")


def _transport_params(init, T=0.04):
    """transport1Dppo.py:40-75 with a short horizon: control_sample_rate 0.01 -> 100 sub-steps, 4 env-steps per episode."""
    dt = 1e-4
    return {"T": T, "dt": dt, "X": 1, "dx": 1e-2, "reward_class": TunedReward1D(int(round(T / dt)), -1e3, 3e2), "normalize": True,
            "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": lambda state: state,
            "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20, "control_sample_rate": 0.01,
            "reset_init_condition_func": init, "reset_recirculation_func": lambda nx: BETA}


def _ns_params(nt=6, n=16):
    """NS2Dppo.py:12-27 on a small grid."""
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    bc = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"],
          "right": ["Dirchilet", "Dirchilet"]}
    return {"T": nt * dt, "dt": dt, "X": 1, "dx": dx, "Y": 1, "dy": dx, "action_dim": 1, "reward_class": NSReward(0.1), "normalize": False,
            "reset_init_condition_func": lambda X: (np.zeros_like(X), np.zeros_like(X), np.zeros_like(X)), "boundary_condition": bc,
            "U_ref": np.zeros((nt, n, n, 2)), "action_ref": 2.0 * np.ones(nt), "maximum_pressure_iteration": 20}



def _five_ids():
    """(id, parameters) of the reference's five registered environments (pde_control_gym/__init__.py:3-18), small horizons."""
    from pde_control_gym.src import BrainTumorReward, TrafficARZReward
    parabolic = dict(_transport_params(lambda nx: np.ones(nx + 1, dtype=np.float32) * 2), dt=1e-5, dx=5e-3, T=4e-3, control_sample_rate=1e-3)
    parabolic["reward_class"] = TunedReward1D(400, -1e3, 3e2)
    parabolic["reset_recirculation_func"] = lambda nx: (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)
    xs = np.linspace(0, 200, 201)
    return [
        ("PDEControlGym-TransportPDE1D", _transport_params(lambda nx: ICS[0])),
        ("PDEControlGym-ReactionDiffusionPDE1D", parabolic),
        ("PDEControlGym-NavierStokes2D", _ns_params()),
        ("PDEControlGym-TrafficPDE1D", dict(T=240, dt=0.25, X=500, dx=10, v_steady=10, ro_steady=0.12, v_max=40, ro_max=0.16, tau=60,
                                           reward_class=TrafficARZReward(), simulation_type="outlet", limit_pde_state_size=True, control_freq=2)),
        ("PDEControlGym-BrainTumor1D", dict(T=50, X=200, dt=1, dx=1, normalize=True, dosage_termination_threshold=0.1, t1_detection_threshold=0.8,
                                           t2_detection_threshold=0.16, D=0.2, rho=0.03, alpha=0.04, alpha_beta_ratio=10, k=1e5,
                                           t1_detection_radius=15, t1_death_radius=35, total_dosage=61.2, verbose=False,
                                           reward_class=BrainTumorReward(),
                                           reset_init_condition_func=lambda X, nx: 0.8 * 1e5 * np.exp(-0.25 * (xs ** 2)))),
    ]


