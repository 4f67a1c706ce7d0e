"""Environments of the reference that are outside this build's hot-path scope.

The names exist so ``from pde_control_gym.src import TrafficPDE1D`` keeps importing; constructing one fails loudly
instead of silently running something else.
"""


class _NotBuilt:
    _what = "this environment"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{self._what} is not part of the MI355X hot-path build yet (SURVEY.md section 8f); "
            "TransportPDE1D, ReactionDiffusionPDE1D and NavierStokes2D are.")


class BrainTumor1D(_NotBuilt):
    _what = "BrainTumor1D (reference environments1d/brain_tumor_env.py)"
