"""Drop-in replacement for lukebhan/PDEControlGym's ``pde_control_gym`` package, MI355X-native engine.

Importing it registers the reference's environment ids (reference pde_control_gym/__init__.py:3-18; the
reference file merges two ``register`` calls at :11-14 and does not parse -- the ids and entry points are kept,
the syntax is fixed).  With gymnasium installed ``gym.make(id, **kwargs)`` works unchanged; without it
``pde_control_gym.make`` offers the same call.
"""
from pde_control_gym._compat import HAVE_GYMNASIUM, make, register

_IDS = {
    "PDEControlGym-TransportPDE1D": "pde_control_gym.src:TransportPDE1D",
    "PDEControlGym-ReactionDiffusionPDE1D": "pde_control_gym.src:ReactionDiffusionPDE1D",
    "PDEControlGym-BrainTumor1D": "pde_control_gym.src:BrainTumor1D",
    "PDEControlGym-TrafficPDE1D": "pde_control_gym.src:TrafficPDE1D",
    "PDEControlGym-NavierStokes2D": "pde_control_gym.src:NavierStokes2D",
    # extension, not in the reference (parity unpinned): nonlinear sibling of TransportPDE1D
    "PDEControlGym-BurgersPDE1D": "pde_control_gym.src.environments1d.burgers:BurgersPDE1D",
}

# gymnasium >= 1.0 also takes a vector entry point: ``gymnasium.make_vec(id, num_envs=N, **params)`` then builds the batched GPU
# environment (pde_control_gym/vector_gymnasium.py) instead of N copies of the single one
_VECTOR = {"PDEControlGym-TransportPDE1D", "PDEControlGym-ReactionDiffusionPDE1D", "PDEControlGym-NavierStokes2D",
           "PDEControlGym-TrafficPDE1D", "PDEControlGym-BurgersPDE1D"}

for _id, _entry in _IDS.items():
    try:
        if _id in _VECTOR:
            try:
                register(id=_id, entry_point=_entry, vector_entry_point="pde_control_gym.vector_gymnasium:vector_" + _id.split("-")[-1])
                continue
            except TypeError:       # gymnasium 0.2x / the fall-back registry: no vector entry points
                pass
        register(id=_id, entry_point=_entry)
    except Exception:  # already registered (module re-import under gymnasium)
        pass

from pde_control_gym.vector import PDEVecEnv, make_vec  # noqa: E402
from pde_control_gym.rollout import DeviceRollout  # noqa: E402
from pdecontrolgym_amd.policy import FusedMLP  # noqa: E402
from pde_control_gym import export  # noqa: E402
from pde_control_gym.vector_gymnasium import GymnasiumVectorAdapter  # noqa: E402

__all__ = ["make", "register", "make_vec", "PDEVecEnv", "DeviceRollout", "FusedMLP", "HAVE_GYMNASIUM", "export", "GymnasiumVectorAdapter"]
