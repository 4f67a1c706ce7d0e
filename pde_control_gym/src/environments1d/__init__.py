from pde_control_gym.src.environments1d.hyperbolic import TransportPDE1D
from pde_control_gym.src.environments1d.parabolic import ReactionDiffusionPDE1D
from pde_control_gym.src.environments1d.traffic_arz_env import TrafficPDE1D
from pde_control_gym.src.environments1d.burgers import BurgersPDE1D
from pde_control_gym.src.environments1d.brain_tumor_env import BrainTumor1D, TherapyWrapper

__all__ = ["TransportPDE1D", "ReactionDiffusionPDE1D", "TrafficPDE1D", "BrainTumor1D", "TherapyWrapper", "BurgersPDE1D"]
