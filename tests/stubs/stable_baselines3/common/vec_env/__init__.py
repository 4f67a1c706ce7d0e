"""The VecEnv contract (public API of stable_baselines3.common.vec_env, SB3 >= 2.0), restated."""
import copy
from abc import ABC, abstractmethod

import numpy as np


class VecEnv(ABC):
    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.reset_infos = [{} for _ in range(num_envs)]
        self._seeds = [None for _ in range(num_envs)]
        self._options = [{} for _ in range(num_envs)]
        render_modes = self.get_attr("render_mode")
        assert all(m == render_modes[0] for m in render_modes), "render_mode mode should be the same for all environments"
        self.render_mode = render_modes[0]
        self.metadata = {"render_modes": [] if self.render_mode is None else [self.render_mode]}

    def _reset_seeds(self):
        self._seeds = [None for _ in range(self.num_envs)]

    def _reset_options(self):
        self._options = [{} for _ in range(self.num_envs)]

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_async(self, actions): ...

    @abstractmethod
    def step_wait(self): ...

    @abstractmethod
    def close(self): ...

    @abstractmethod
    def get_attr(self, attr_name, indices=None): ...

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None): ...

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...

    @abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None): ...

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed=None):
        if seed is None:
            seed = int(np.random.randint(0, np.iinfo(np.uint32).max, dtype=np.uint32))
        self._seeds = [seed + idx for idx in range(self.num_envs)]
        return self._seeds

    def set_options(self, options=None):
        if options is None:
            options = {}
        self._options = copy.deepcopy([options] * self.num_envs) if isinstance(options, dict) else copy.deepcopy(options)

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def getattr_depth_check(self, name, already_found):
        return f"{type(self).__module__}.{type(self).__name__}" if hasattr(self, name) and already_found else None

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices


class VecEnvWrapper(VecEnv):
    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(venv.num_envs, observation_space or venv.observation_space, action_space or venv.action_space)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def close(self):
        return self.venv.close()

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def set_options(self, options=None):
        return self.venv.set_options(options)

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self.venv.set_attr(attr_name, value, indices)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return self.venv.env_method(method_name, *method_args, indices=indices, **method_kwargs)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return self.venv.env_is_wrapped(wrapper_class, indices=indices)


class VecMonitor(VecEnvWrapper):
    """Episode statistics: on a finished episode ``infos[i]["episode"] = {"r": return, "l": length}`` (the info dict is copied)."""

    def __init__(self, venv):
        super().__init__(venv)
        self.episode_returns = np.zeros(self.num_envs, dtype=np.float32)
        self.episode_lengths = np.zeros(self.num_envs, dtype=np.int32)

    def reset(self):
        obs = self.venv.reset()
        self.episode_returns[:] = 0
        self.episode_lengths[:] = 0
        return obs

    def step_wait(self):
        obs, rewards, dones, infos = self.venv.step_wait()
        self.episode_returns += rewards
        self.episode_lengths += 1
        new_infos = list(infos[:])
        for i in range(len(dones)):
            if dones[i]:
                info = infos[i].copy()
                info["episode"] = {"r": float(self.episode_returns[i]), "l": int(self.episode_lengths[i])}
                self.episode_returns[i] = 0
                self.episode_lengths[i] = 0
                new_infos[i] = info
        return obs, rewards, dones, new_infos


def unwrap_vec_wrapper(env, vec_wrapper_class):
    env_tmp = env
    while isinstance(env_tmp, VecEnvWrapper):
        if isinstance(env_tmp, vec_wrapper_class):
            return env_tmp
        env_tmp = env_tmp.venv
    return None


def is_vecenv_wrapped(env, vec_wrapper_class):
    return unwrap_vec_wrapper(env, vec_wrapper_class) is not None
