"""The type gate every SB3 algorithm applies to its ``env`` argument (BaseAlgorithm._wrap_env), restated."""
import gymnasium

from stable_baselines3.common.vec_env import VecEnv


class WouldWrapInDummyVecEnv(Exception):
    """Raised by this stand-in where SB3 would go on to wrap a single gymnasium.Env into a DummyVecEnv of one."""


def _patch_env(env):
    if isinstance(env, gymnasium.Env):
        return env
    raise ValueError(f"The environment is of type {type(env)}, not a Gymnasium environment. "
                     "In this case, we expect OpenAI Gym to be installed and the environment to be an OpenAI Gym environment.")


def _wrap_env(env, verbose=0, monitor_wrapper=True):
    if not isinstance(env, VecEnv):
        env = _patch_env(env)
        raise WouldWrapInDummyVecEnv(type(env).__name__)
    assert not isinstance(env.observation_space, gymnasium.spaces.Dict)
    return env
