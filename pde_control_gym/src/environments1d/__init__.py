from pde_control_gym.src.environments1d.hyperbolic import TransportPDE1D
from pde_control_gym.src.environments1d.parabolic import ReactionDiffusionPDE1D
from pde_control_gym.src.environments1d.not_built import TrafficPDE1D, BrainTumor1D

__all__ = ["TransportPDE1D", "ReactionDiffusionPDE1D", "TrafficPDE1D", "BrainTumor1D"]
