"""Same import surface as the reference's pde_control_gym/src/__init__.py:1-5."""
from pde_control_gym.src.environments1d import TransportPDE1D, ReactionDiffusionPDE1D, TrafficPDE1D, BrainTumor1D, TherapyWrapper
from pde_control_gym.src.environments2d import NavierStokes2D
from pde_control_gym.src.rewards import BaseReward, NormReward, TunedReward1D, NSReward, TrafficARZReward, BrainTumorReward

__all__ = ["TransportPDE1D", "ReactionDiffusionPDE1D", "NavierStokes2D", "BaseReward", "NormReward", "TunedReward1D",
           "NSReward", "TrafficPDE1D", "TrafficARZReward", "BrainTumor1D", "TherapyWrapper", "BrainTumorReward"]
