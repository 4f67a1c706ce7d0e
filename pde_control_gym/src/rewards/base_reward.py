"""Reward extension point (reference rewards/base_reward.py:5-32): subclass and override ``reward``.

A user subclass is honoured by every environment through the history-view path (the step kernel then
evaluates no reward; ``reward`` is called on the host once per step with a lazy view of the trajectory).
The shipped rewards (TunedReward1D, NormReward, NSReward) are additionally recognised by type and evaluated
inside the step kernel.
"""
from abc import ABC, abstractmethod


class BaseReward(ABC):
    @abstractmethod
    def reward(self, uVec=None, time_index=None, terminate=None, truncate=None, action=None):
        """uVec: trajectory (row t = state at sub-step t); time_index: row to score."""

    def reset(self):
        """Hook for stateful rewards.  (The reference declares it but no environment ever calls it; the same here.)"""
        pass
