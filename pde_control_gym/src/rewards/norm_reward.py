"""NormReward -- L1 / L2 / Linf state-norm rewards.

The reference class (rewards/norm_reward.py:19-73) cannot run: its first call compares an ndarray with
``None`` (ValueError) and it divides by an undefined ``norm_coeff``.  This is a working implementation of the
documented intent (reference docs/source/utils/preimplementedrewards.rst:10-14): PARITY UNPINNED.

  horizon="temporal"      -||u_t||                      (evaluated inside the step kernel)
  horizon="differential"  ||u_t - u_{t-1}||  (t > 0)    (evaluated inside the step kernel: PDEGYM_HORIZON_DIFFERENTIAL)
  horizon="t-horizon"     -mean of the last k norms     (evaluated inside the step kernel for k <= 128: PDEGYM_HORIZON_T)
"""
import numpy as np

from pde_control_gym.src.rewards.base_reward import BaseReward

_ORD = {"1": 1, "2": 2, "inf": np.inf}


class NormReward(BaseReward):
    def __init__(self, nt: int = None, norm: str = "2", horizon: str = "temporal", truncate_penalty: float = -1e-4,
                 terminate_reward: float = 1e2, t_horizon_length: int = 5, *extras):
        if nt is None:
            raise Exception("Number of simulation steps must be specified in the NormReward class.")
        if str(norm) not in _ORD:
            raise Exception("Invalid norm parameter. Please use '1', '2' or 'inf'.")
        if horizon not in ("temporal", "differential", "t-horizon"):
            raise Exception("Invalid horizon parameter. Please use 'temporal', 'differential' or 't-horizon'.")
        self.nt = nt
        self.norm = str(norm)
        self.horizon = horizon
        self.truncate_penalty = truncate_penalty
        self.terminate_reward = terminate_reward
        self.t_horizon_length = t_horizon_length

    def reward(self, uVec=None, time_index=None, terminate=None, truncate=None, action=None):
        if uVec is None:
            raise Exception("Class NormReward attempted to call reward function and recieved a None vector to compute on")
        if time_index is None:
            raise Exception("Class NormReward attempted to call reward fucntion and recieved a None time_index parameter to identify the reward step")
        if terminate:
            return self.terminate_reward
        if truncate:
            return self.truncate_penalty * (self.nt - time_index)
        o = _ORD[self.norm]
        if self.horizon == "temporal":
            return -np.linalg.norm(uVec[time_index], ord=o)
        if self.horizon == "differential":
            if time_index > 0:
                return np.linalg.norm(uVec[time_index] - uVec[time_index - 1], ord=o)
            return -np.linalg.norm(uVec[time_index], ord=o)
        k = min(self.t_horizon_length, time_index + 1)
        return -sum(np.linalg.norm(uVec[time_index - i], ord=o) for i in range(k)) / k
