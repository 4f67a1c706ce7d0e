"""TransportPDE1D -- u_t = u_x + beta(x) u(0,t) with boundary control at x = X
(interface of the reference's environments1d/hyperbolic.py:25-227).

First-order upwind, explicit Euler; the non-local recirculation term, the five sensing modes, the
Dirichlet ("Dirchilet" in the reference's spelling, kept) / Neumann control types and sub-stepping are
executed by pdegym_transport_step on the GPU.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from pde_control_gym._compat import spaces
from pde_control_gym.src.environments1d.base_env_1d import PDEEnv1D, validate_1d_options


class TransportPDE1D(PDEEnv1D):
    """:param sensing_noise_func: applied to every observation before it is returned.
    :param reset_init_condition_func: ``f(nx) -> u(x, 0)`` (length nx), called at every reset.
    :param reset_recirculation_func: ``f(nx) -> beta(x)`` (length nx), called at every reset.
    :param sensing_loc: "full" | "collocated" | "opposite".
    :param control_type: "Dirchilet" | "Neumann" (at x = X).
    :param sensing_type: "Dirchilet" | "Neumann" (used when ``sensing_loc == "opposite"``).
    :param limit_pde_state_size: truncate when ||u||_2 >= ``max_state_value``.
    :param max_control_value: action scale when ``normalize`` is on.
    :param control_sample_rate: seconds of simulated time per ``step`` call.
    Extra (not in the reference): ``device`` ("cuda"), ``record_history`` (keep ``env.u`` on device)."""

    _kind = "transport"

    def __init__(self, sensing_noise_func: Callable[[np.ndarray], np.ndarray],
                 reset_init_condition_func: Callable[[int], np.ndarray],
                 reset_recirculation_func: Callable[[int], np.ndarray],
                 sensing_loc: str = "full", control_type: str = "Dirchilet", sensing_type: str = "Dirchilet",
                 limit_pde_state_size: bool = False, max_state_value: float = 1e10, max_control_value: float = 20,
                 control_sample_rate: float = 0.1, device="cuda", record_history: bool = True, backend=None, **kwargs):
        super().__init__(**kwargs)
        self.sensing_noise_func = sensing_noise_func
        self.reset_init_condition_func = reset_init_condition_func
        self.reset_recirculation_func = reset_recirculation_func
        self.sensing_loc, self.control_type, self.sensing_type = sensing_loc, control_type, sensing_type
        self.limit_pde_state_size = limit_pde_state_size
        self.max_state_value = max_state_value
        self.max_control_value = max_control_value
        self.control_sample_rate = control_sample_rate
        validate_1d_options(self._kind, sensing_loc, control_type, sensing_type)
        dim = self.nx if sensing_loc == "full" else 1
        self.observation_space = spaces.Box(np.full(dim, -self.max_state_value, dtype="float32"),
                                            np.full(dim, self.max_state_value, dtype="float32"))
        self._build_engine(device, record_history, backend)
