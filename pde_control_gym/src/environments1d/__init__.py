from pde_control_gym.src.environments1d.hyperbolic import TransportPDE1D
from pde_control_gym.src.environments1d.parabolic import ReactionDiffusionPDE1D
from pde_control_gym.src.environments1d.traffic_arz_env import TrafficPDE1D
from pde_control_gym.src.environments1d.not_built import BrainTumor1D

__all__ = ["TransportPDE1D", "ReactionDiffusionPDE1D", "TrafficPDE1D", "BrainTumor1D"]
