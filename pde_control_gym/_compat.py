"""gymnasium is optional: neither gymnasium nor stable-baselines3 ships in the MI355X image.

When gymnasium is importable its ``Env``/``spaces.Box``/``register``/``make`` are used, so
``gym.make("PDEControlGym-TransportPDE1D", **kwargs)`` and SB3 work unchanged.  Otherwise a minimal stand-in
provides the same attributes the environments touch (metadata only -- no arithmetic lives here).
"""
from __future__ import annotations

import importlib

import numpy as np

try:  # pragma: no cover - depends on the host image
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    from gymnasium.envs.registration import register as _register
    HAVE_GYMNASIUM = True
except Exception:  # ImportError or a broken install
    _gym = None
    HAVE_GYMNASIUM = False


class _Box:
    """Subset of gymnasium.spaces.Box: low/high/shape/dtype, sample(), contains()."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        self.shape = tuple(shape)
        self.dtype = dtype
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class _Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None

    def __init__(self):
        pass

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass

    def render(self):
        return None


class _Wrapper(_Env):
    """Subset of gymnasium.Wrapper: holds ``env``, forwards reset/step/attributes, ``unwrapped`` reaches the base env."""

    def __init__(self, env):
        self.env = env

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()


class _Spaces:
    Box = _Box


_REGISTRY = {}


def _fallback_register(id, entry_point, **kwargs):
    _REGISTRY[id] = (entry_point, kwargs)


def _fallback_make(id, **kwargs):
    if id not in _REGISTRY:
        raise KeyError(f"No registered env with id: {id}")
    entry_point, defaults = _REGISTRY[id]
    if isinstance(entry_point, str):
        mod, attr = entry_point.split(":")
        entry_point = getattr(importlib.import_module(mod), attr)
    return entry_point(**{**defaults.get("kwargs", {}), **kwargs})


if HAVE_GYMNASIUM:  # pragma: no cover
    Env, spaces, register, make, Wrapper = _gym.Env, _spaces, _register, _gym.make, _gym.Wrapper
else:
    Env, spaces, register, make, Wrapper = _Env, _Spaces, _fallback_register, _fallback_make, _Wrapper
