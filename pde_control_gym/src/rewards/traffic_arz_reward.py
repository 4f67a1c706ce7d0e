"""TrafficARZReward belongs to the TrafficPDE1D environment, which is outside this build's hot-path scope
(SURVEY.md section 8f rank 2).  The name is kept importable; using it fails loudly."""
from pde_control_gym.src.rewards.base_reward import BaseReward


class TrafficARZReward(BaseReward):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TrafficARZReward / TrafficPDE1D are not part of the MI355X hot-path build yet")

    def reward(self, *args, **kwargs):  # pragma: no cover
        raise NotImplementedError
