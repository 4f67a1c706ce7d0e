"""TunedReward1D -- the reward every 1D example of the reference uses (rewards/tuned_reward_1d.py:17-40).

Inside ``env.step`` this reward is evaluated by the HIP step kernel from three streaming scalars per
instance (||u_t||, ||u_{t-100}||, running sum of |u[tau,-1]|); see pdecontrolgym_amd/csrc/pdegym_1d.hip.
``reward()`` below is the host-side definition for callers that hold a trajectory array themselves.
"""
import numpy as np

from pde_control_gym.src.rewards.base_reward import BaseReward


class TunedReward1D(BaseReward):
    """:param nt: number of simulation steps of the episode (required).
    :param truncate_penalty: per-remaining-step penalty when the episode is truncated (default -1e-4).
    :param terminate_reward: bonus for reaching the end of the horizon with a small state (default 1e2)."""

    #: rows the shaping term looks back; the reference hard-wires int(1/0.01) through a default kwarg the
    #: environments never override (tuned_reward_1d.py:25,40)
    LOOKBACK = 100

    def __init__(self, nt: int, truncate_penalty: float = -1e-4, terminate_reward: float = 1e2):
        if nt is None:
            raise Exception("Number of simulation steps must be specified in the NormReward class.")
        self.nt = nt
        self.truncate_penalty = truncate_penalty
        self.terminate_reward = terminate_reward

    def reward(self, uVec=None, time_index=None, terminate=None, truncate=None, action=None, control_sample_rate=0.01):
        now = np.linalg.norm(uVec[time_index])
        if terminate and now < 20:
            return self.terminate_reward - np.sum(abs(uVec[:, -1])) / 1000 - now
        if truncate:
            return self.truncate_penalty * (self.nt - time_index)
        return np.linalg.norm(uVec[time_index - int(1 / control_sample_rate)]) - now
