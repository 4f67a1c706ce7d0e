"""TrafficARZReward (reference rewards/traffic_arz_reward.py:5-22): minus the relative L2 deviation of velocity and
density from the desired steady state.  ``TrafficPDE1D.step`` evaluates it inside the step kernel; ``reward()`` is the
host-side definition for caller-held arrays."""
import numpy as np

from pde_control_gym.src.rewards.base_reward import BaseReward


class TrafficARZReward(BaseReward):
    def reward(self, v_desired: float, r_desired: float, v: np.ndarray, r: np.ndarray):
        dev_v = np.linalg.norm(v - v_desired, ord=None) / v_desired
        dev_r = np.linalg.norm(r - r_desired, ord=None) / r_desired
        return -(dev_v + dev_r)
