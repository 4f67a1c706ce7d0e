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


def _fallback_register(id, entry_point, vector_entry_point=None, **kwargs):
    _REGISTRY[id] = (entry_point, kwargs)
    if vector_entry_point is not None:
        _VECTOR_REGISTRY[id] = vector_entry_point


_VECTOR_REGISTRY = {}


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


# ---- the batched faces' base classes ------------------------------------------------------------------------------------
# Stable-Baselines3 decides with ``isinstance(env, VecEnv)`` whether an environment is already vectorised
# (BaseAlgorithm._wrap_env); anything else is treated as ONE gymnasium.Env and rejected if it is not one.  So the batched
# faces must BE stable_baselines3.common.vec_env.VecEnv / gymnasium.vector.VectorEnv subclasses whenever those packages are
# importable -- the caller to match is ``PPO("MlpPolicy", env)`` (reference examples/transportPDE/transport1Dppo.py:77-90).
# Without the packages, stand-ins with the same constructor contract and helper methods keep the behaviour identical.

class _VecEnv:
    """What stable_baselines3.common.vec_env.VecEnv gives its subclasses (public contract, SB3 >= 2.0): the constructor
    ``(num_envs, observation_space, action_space)`` fills ``reset_infos`` / ``_seeds`` / ``_options`` and reads the common
    ``render_mode`` through ``get_attr``; ``step`` = ``step_async`` + ``step_wait``; ``seed`` / ``set_options`` park their
    arguments for the next ``reset``."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.reset_infos = [{} for _ in range(num_envs)]
        self._seeds = [None for _ in range(num_envs)]
        self._options = [{} for _ in range(num_envs)]
        try:
            modes = self.get_attr("render_mode")
        except AttributeError:
            modes = [None for _ in range(num_envs)]
        self.render_mode = modes[0] if modes else None
        self.metadata = {"render_modes": [] if self.render_mode is None else [self.render_mode]}

    def _reset_seeds(self):
        self._seeds = [None for _ in range(self.num_envs)]

    def _reset_options(self):
        self._options = [{} for _ in range(self.num_envs)]

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        if seed is None:
            seed = int(np.random.randint(0, np.iinfo(np.uint32).max, dtype=np.uint32))
        self._seeds = [seed + i for i in range(self.num_envs)]
        return self._seeds

    def set_options(self, options=None):
        import copy
        if options is None:
            options = {}
        self._options = copy.deepcopy([options] * self.num_envs if isinstance(options, dict) else options)

    def get_images(self):
        raise NotImplementedError

    def render(self, mode=None):
        return None

    @property
    def unwrapped(self):
        return self

    def getattr_depth_check(self, name, already_found):
        return f"{type(self).__module__}.{type(self).__name__}" if hasattr(self, name) and already_found else None

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices


class _VectorEnv:
    """What gymnasium.vector.VectorEnv gives its subclasses (gymnasium >= 1.0: class attributes, no constructor arguments)."""
    metadata = {}
    spec = None
    render_mode = None
    closed = False
    num_envs = 0

    @property
    def unwrapped(self):
        return self

    def render(self):
        raise NotImplementedError

    def close_extras(self, **kwargs):
        pass

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True


def _fallback_batch_space(space, n):
    return _Box(np.stack([space.low] * n), np.stack([space.high] * n), dtype=space.dtype)


try:  # pragma: no cover - depends on the host image
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
    HAVE_SB3 = True
except Exception:
    _SB3VecEnv, HAVE_SB3 = None, False

_GymVectorEnv, batch_space, AUTORESET_SAME_STEP = None, _fallback_batch_space, "same_step"
if HAVE_GYMNASIUM:  # pragma: no cover
    try:
        from gymnasium.vector import VectorEnv as _GymVectorEnv
        from gymnasium.vector.utils import batch_space
    except Exception:
        _GymVectorEnv, batch_space = None, _fallback_batch_space
    try:
        from gymnasium.vector import AutoresetMode as _AutoresetMode      # gymnasium >= 1.0
        AUTORESET_SAME_STEP = _AutoresetMode.SAME_STEP
    except Exception:
        pass

VecEnv = _SB3VecEnv if HAVE_SB3 else _VecEnv
VectorEnv = _GymVectorEnv if _GymVectorEnv is not None else _VectorEnv
HAVE_GYMNASIUM_VECTOR = _GymVectorEnv is not None
