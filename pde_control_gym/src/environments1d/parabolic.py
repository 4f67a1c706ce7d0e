"""ReactionDiffusionPDE1D -- u_t = u_xx + lambda(x) u, u(0,t) = 0, boundary control at x = X
(interface of the reference's environments1d/parabolic.py:25-221).

FTCS three-point Laplacian plus the local reaction term on nx+1 nodes (ghost point), executed by
pdegym_parabolic_step on the GPU.  Stability (dt/dx**2 <= 0.5) is the caller's business, as in the reference.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from pde_control_gym._compat import spaces
from pde_control_gym.src.environments1d.base_env_1d import PDEEnv1D, validate_1d_options


class ReactionDiffusionPDE1D(PDEEnv1D):
    """Same keywords as :class:`TransportPDE1D`; the reset callbacks still receive ``nx`` but must return
    ``nx+1`` values, and Dirichlet sensing at the opposite end is rejected (u(0,t) is identically 0)."""

    _kind = "parabolic"

    def __init__(self, sensing_noise_func: Callable[[np.ndarray], np.ndarray],
                 reset_init_condition_func: Callable[[int], np.ndarray],
                 reset_recirculation_func: Callable[[int], np.ndarray],
                 sensing_loc: str = "full", control_type: str = "Dirchilet", sensing_type: str = "Dirchilet",
                 limit_pde_state_size: bool = False, max_state_value: float = 1e10, max_control_value: float = 20,
                 control_sample_rate: float = 1e-4, device="cuda", record_history: bool = True, backend=None, **kwargs):
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
        dim = self.nx + 1 if sensing_loc == "full" else 1
        self.observation_space = spaces.Box(np.full(dim, -self.max_state_value, dtype="float32"),
                                            np.full(dim, self.max_state_value, dtype="float32"))
        self._build_engine(device, record_history, backend)
