"""PDEEnv2D -- base class of the 2D environments (interface of the reference's
environments2d/base_env_2d.py:10-63): grid sizes, meshgrid, observation/action spaces, ``normalize``."""
from __future__ import annotations

from abc import abstractmethod

import numpy as np

from pde_control_gym._compat import Env, spaces
from pde_control_gym.src.rewards import BaseReward


class PDEEnv2D(Env):
    """:param T, dt: horizon and time step.  :param X, dx, Y, dy: domain and grid spacing.
    :param action_dim: length of the action vector.  :param reward_class: a BaseReward instance.
    :param normalize: builds the same ``normalize`` callable as the 1D base class (the Navier-Stokes env, like
    the reference's, never applies it: actions are used raw)."""

    def __init__(self, T: float, dt: float, X: float, dx: float, Y: float, dy: float, action_dim: int,
                 reward_class: BaseReward, normalize: bool = False):
        super().__init__()
        self.nt = int(round(T / dt))
        self.nx = int(round(X / dx + 1))
        self.ny = int(round(Y / dy + 1))
        self.dx, self.dy, self.dt = dx, dy, dt
        self.x = np.linspace(0, X, self.nx)
        self.y = np.linspace(0, Y, self.ny)
        self.X, self.Y = np.meshgrid(self.x, self.y)
        self.observation_space = spaces.Box(np.full((self.nx, self.ny, 2), -np.inf, dtype="float32"),
                                            np.full((self.nx, self.ny, 2), np.inf, dtype="float32"))
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(action_dim,), dtype=np.float32)
        self.action_dim = action_dim
        if normalize:
            self.normalize = lambda action, max_value: (action + 1) * max_value - max_value
        else:
            self.normalize = lambda action, max_value: action
        self.U = np.zeros((self.nt, self.nx, self.ny, 2))
        self.time_index = 0
        self.reward_class = reward_class

    @abstractmethod
    def step(self, action):
        pass

    @abstractmethod
    def reset(self, seed=None, options=None):
        pass
