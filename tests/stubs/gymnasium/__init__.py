"""Stand-in for gymnasium (see tests/stubs/README.md): the public names pde_control_gym touches, no more."""
import importlib

from gymnasium import spaces, vector  # noqa: F401
from gymnasium.envs.registration import register, registry, vector_registry  # noqa: F401

__version__ = "0.0+contract-stub"


class Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    action_space = None
    observation_space = None

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def render(self):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()


def make(id, **kwargs):
    if id not in registry:
        raise KeyError(f"No registered env with id: {id}")
    entry_point, defaults = registry[id]
    if isinstance(entry_point, str):
        mod, attr = entry_point.split(":")
        entry_point = getattr(importlib.import_module(mod), attr)
    return entry_point(**{**defaults, **kwargs})


def make_vec(id, num_envs=1, vectorization_mode=None, vector_kwargs=None, wrappers=None, **kwargs):
    """gymnasium >= 1.0: a spec with a vector entry point is built through it (vectorization_mode "vector_entry_point")."""
    if id not in registry:
        raise KeyError(f"No registered env with id: {id}")
    if id not in vector_registry or vectorization_mode in ("sync", "async"):
        raise NotImplementedError("the stand-in only builds native vector environments")
    entry = vector_registry[id]
    if isinstance(entry, str):
        mod, attr = entry.split(":")
        entry = getattr(importlib.import_module(mod), attr)
    return entry(num_envs=num_envs, **{**registry[id][1], **kwargs})
