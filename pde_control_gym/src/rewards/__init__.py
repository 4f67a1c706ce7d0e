from pde_control_gym.src.rewards.norm_reward import NormReward
from pde_control_gym.src.rewards.tuned_reward_1d import TunedReward1D
from pde_control_gym.src.rewards.base_reward import BaseReward
from pde_control_gym.src.rewards.ns_reward import NSReward
from pde_control_gym.src.rewards.traffic_arz_reward import TrafficARZReward
from pde_control_gym.src.rewards.brain_tumor_reward import BrainTumorReward

__all__ = ["NormReward", "TunedReward1D", "BaseReward", "NSReward", "TrafficARZReward", "BrainTumorReward"]
