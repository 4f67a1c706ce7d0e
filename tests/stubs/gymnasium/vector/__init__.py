from enum import Enum

from gymnasium.vector import utils  # noqa: F401


class AutoresetMode(Enum):
    NEXT_STEP = "NextStep"
    SAME_STEP = "SameStep"
    DISABLED = "Disabled"


class VectorEnv:
    """gymnasium >= 1.0: no constructor arguments; subclasses set the attributes."""
    metadata = {}
    spec = None
    render_mode = None
    closed = False
    observation_space = None
    action_space = None
    single_observation_space = None
    single_action_space = None
    num_envs = 0

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, actions):
        raise NotImplementedError

    def render(self):
        raise NotImplementedError

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True

    def close_extras(self, **kwargs):
        pass

    @property
    def unwrapped(self):
        return self
