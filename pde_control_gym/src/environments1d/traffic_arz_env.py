"""TrafficPDE1D -- Aw-Rascle-Zhang freeway model with flux boundary control at the inlet and/or the outlet
(interface of the reference's environments1d/traffic_arz_env.py:8-280).

State (density ``r``, relative flow ``y``) lives on the GPU in float64; one ``step`` = ``control_freq`` two-step
Lax-Wendroff sub-steps with relaxation in one kernel launch (pdecontrolgym_amd/csrc/pdegym_traffic.hip), bit-identical
to the reference's NumPy arithmetic.  Kept on purpose: simulated time advances by ``dt`` once per ``step`` whatever
``control_freq`` is; ``terminate()`` compares seconds with ``T/dt``; 'outlet-train' redraws the steady state at every
reset while its action bounds stay those of the construction-time draw; ``reset`` returns raw (r, v) for every variant.
``simulation_type='inlet-train'`` is rejected: in the reference it raises AttributeError on the first step (its inlet
flux is never assigned).
"""
from __future__ import annotations

import random

import numpy as np

from pde_control_gym._compat import spaces
from pde_control_gym.src.environments1d.base_env_1d import PDEEnv1D


class TrafficPDE1D(PDEEnv1D):
    """:param simulation_type: 'inlet' | 'outlet' | 'both' | 'outlet-train'.
    :param v_steady, ro_steady: desired steady state (must satisfy v = v_max (1 - ro/ro_max)).
    :param v_max, ro_max, tau: maximum velocity / density, relaxation time.
    :param limit_pde_state_size: truncate when v > v_max or r > ro_max anywhere.
    :param control_freq: PDE sub-steps per ``step`` call.  Extra: ``device``."""

    def __init__(self, simulation_type: str = "inlet", v_steady: float = 10, ro_steady: float = 0.12, v_max: float = 40,
                 ro_max: float = 0.16, tau: float = 60, limit_pde_state_size: bool = False, control_freq: int = 1,
                 device="cuda", backend=None, **kwargs):
        super().__init__(**kwargs)
        from pdecontrolgym_amd.batch_traffic import TrafficBatch
        self.simulation_type = simulation_type
        self.vm, self.rm, self.qm, self.tau = v_max, ro_max, v_max * ro_max / 4, tau
        self.limit_pde_state_size = limit_pde_state_size
        assert (isinstance(control_freq, int) and control_freq >= 1), \
            f"control_freq must be a positive integer (got {control_freq} of type {type(control_freq).__name__})"
        self.control_freq = control_freq
        if simulation_type == "inlet-train":
            raise ValueError("simulation_type 'inlet-train' cannot step in the reference (q_inlet is never assigned)")
        if simulation_type not in ("outlet", "inlet", "both", "outlet-train"):
            raise ValueError("Invalid simulation type")
        if simulation_type in ("inlet", "outlet", "both"):
            if v_steady != TrafficPDE1D.Veq(v_max, ro_max, ro_steady):
                raise ValueError("The steady state velocity and density do not satisfy the equilibrium condition. Check the values "
                                 "of v_steady and ro_steady and ensure that they obey v_steady = v_max(1 - ro_steady/v_max).")
            self.vs, self.rs = v_steady, ro_steady
            self.qs = v_steady * ro_steady
            self.ps = self.vm / self.rm * self.qs / self.vs
        else:
            self.rs = {0: 0.115, 1: 0.12, 2: 0.125}[random.randint(0, 2)]
            self.vs = TrafficPDE1D.Veq(self.vm, self.rm, self.rs)
            self.qs = self.rs * self.vs
        self._core = TrafficBatch(self.T, self.dt, self.X, self.dx, simulation_type, v_max, ro_max, tau, limit_pde_state_size,
                                  control_freq, num_envs=1, device=device, backend=backend)
        self.L, self.M = self.X, self._core.M
        self.qs_input = np.linspace(self.qs / 2, 2 * self.qs, 40)
        self._core.set_action_bounds([self.qs])                 # action space fixed at construction (:97-100)
        if simulation_type == "outlet-train":
            self.observation_space = spaces.Box(low=-10, high=10, shape=(2 * self.M,), dtype=np.float64)
        else:
            self.observation_space = spaces.Box(low=0, high=40, shape=(2 * self.M,), dtype=np.float64)
        self.action_space = spaces.Box(dtype=np.float64, low=self.qs * 0.8, high=1.2 * self.qs,
                                       shape=(2,) if simulation_type == "both" else (1,))
        self.info = dict()
        # one launch, ONE device-to-host copy and one synchronisation per step(): the command is read by the kernel from pinned host
        # memory in place; observation, fields, clock, reward and flags share one allocation (TrafficBatch.host_pack)
        from pdecontrolgym_amd.hostio import HostFetch, PinnedInputs
        self._fetch, self._pins, self._views = HostFetch(self._core.device), PinnedInputs(self._core.device), None
        self._core.reset([self.rs])
        self._load_state()
        self.info["V"] = self.v

    def _load_state(self):
        """Observation, fields, clock, reward and flags of the last launch: ONE device-to-host copy of the engine's pack."""
        core = self._core
        raw = self._fetch([core.host_pack])[0]
        if self._views is None or self._views[0] is not raw:
            self._views = (raw, core.pack_layout.numpy_views(raw))
        v = self._views[1]
        self.r = v["r"][0].reshape(self.M, 1).copy()
        self.y = v["y"][0].reshape(self.M, 1).copy()
        self.v = self.y / self.r + TrafficPDE1D.Veq(self.vm, self.rm, self.r)
        self.time_index = float(v["time"][0])
        return v["obs1" if core.t["obs"] is core._obs[1] else "obs0"][0].copy(), v

    def terminate(self):
        return bool(self._done_flag)

    def truncate(self):
        return bool(self._trunc_flag)

    def step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        import torch
        self._core.step(self._pins("action", a[None, : self._core.action_dim], torch.float64))
        o, v = self._load_state()
        self._done_flag, self._trunc_flag = bool(v["done"][0]), bool(v["truncated"][0])
        reward = float(v["reward"][0])
        from pde_control_gym.src.rewards import TrafficARZReward
        if type(self.reward_class) is not TrafficARZReward:
            # a user reward class (docs/source/utils/customrewards.rst): called as the reference does (:228); outside
            # 'outlet-train' the episode also ends when THIS reward exceeds -0.00023 (:233)
            reward = self.reward_class.reward(self.vs, self.rs, self.v, self.r)
            if self.simulation_type != "outlet-train":
                self._done_flag = bool(self.time_index == 0 or reward > -0.00023)    # terminate() rewinds the clock when it fires
        return o, reward, self._done_flag, self._trunc_flag, self.info

    def reset(self, seed=None, options=None):
        if self.simulation_type == "outlet-train":               # stochastic reset (:249-253)
            self.rs = {0: 0.115, 1: 0.12, 2: 0.125}[random.randint(0, 2)]
            self.vs = TrafficPDE1D.Veq(self.vm, self.rm, self.rs)
            self.qs = self.rs * self.vs
        self._core.reset([self.rs])
        o, _ = self._load_state()
        self._done_flag = self._trunc_flag = False
        return o, {}

    @staticmethod
    def Veq(vm, rm, rho):
        return vm * (1 - rho / rm)

    @staticmethod
    def F_r(vm, rm, rho, y):
        return y + rho * TrafficPDE1D.Veq(vm, rm, rho)

    @staticmethod
    def F_y(vm, rm, rho, y):
        return y * (y / rho + TrafficPDE1D.Veq(vm, rm, rho))
