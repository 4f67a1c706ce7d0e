"""NSReward -- trajectory-tracking reward of the Navier-Stokes env (reference rewards/ns_reward.py:15-28).

``env.step`` evaluates it inside the NS2D step kernel (block reduction of ||U_t - U_ref,t||^2);
``reward()`` is the host-side definition for callers holding arrays.
"""
import numpy as np

from pde_control_gym.src.rewards.base_reward import BaseReward


class NSReward(BaseReward):
    """:param gamma: weight of the action cost."""

    def __init__(self, gamma: float = 0.1):
        self.gamma = gamma

    def reward(self, uVec=None, time_index=None, U_ref=None, action=None, action_ref=None):
        track = np.linalg.norm(uVec[time_index] - U_ref[time_index]) ** 2
        effort = np.linalg.norm(action - action_ref[time_index]) ** 2
        return - 1 / 2 * track / uVec.shape[1] / uVec.shape[2] - self.gamma / 2 * effort
