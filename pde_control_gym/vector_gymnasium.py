"""gymnasium.vector-style face over the batched environments (SURVEY.md section 8f rank 1 names both the SB3 ``VecEnv`` and the
``gymnasium.vector`` conventions).

``GymnasiumVectorAdapter(venv)`` wraps a ``PDEVecEnv`` / ``TumorVecEnv`` and speaks the gymnasium vector API:

    obs, infos = venv.reset(seed=None, options=None)
    obs, rewards, terminations, truncations, infos = venv.step(actions)

with same-step auto-reset: a finished sub-environment already returns the first observation of its next episode, and the
observation its episode ended on is in ``infos["final_obs"][i]`` (mask ``infos["_final_obs"]``; gymnasium >= 1.0's
``AutoresetMode.SAME_STEP`` layout, with the matching ``final_info``) as well as ``infos["final_observation"][i]`` (gymnasium
0.29's ``SyncVectorEnv`` layout).  Nothing is computed here; it only re-shapes what the SB3 face returns, so the two
conventions cannot drift apart.

The class IS a ``gymnasium.vector.VectorEnv`` whenever gymnasium is importable (wrappers and trainers test with ``isinstance``);
``observation_space`` / ``action_space`` are the batched spaces, ``single_*`` the per-environment ones, as that class documents.
"""
from __future__ import annotations

import numpy as np

from pde_control_gym._compat import AUTORESET_SAME_STEP, VectorEnv, batch_space


class GymnasiumVectorAdapter(VectorEnv):
    metadata = {"autoreset_mode": AUTORESET_SAME_STEP, "render_modes": []}

    def __init__(self, venv):
        try:                        # gymnasium >= 1.0 (and the stand-in): no constructor arguments, attributes set below
            VectorEnv.__init__(self)
        except TypeError:           # gymnasium 0.2x: VectorEnv(num_envs, observation_space, action_space) batches the spaces itself
            VectorEnv.__init__(self, venv.num_envs, venv.observation_space, venv.action_space)
        self.venv = venv
        self.num_envs = venv.num_envs
        self.single_observation_space = venv.observation_space
        self.single_action_space = venv.action_space
        self.observation_space = batch_space(venv.observation_space, self.num_envs)
        self.action_space = batch_space(venv.action_space, self.num_envs)
        self.render_mode = None
        self.closed = False

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        # the reference's reset(seed, options) ignores both (hyperbolic.py:196-227)
        return self.venv.reset(), {}

    def step(self, actions):
        obs, rew, dones, infos = self.venv.step(actions)
        term = np.zeros(self.num_envs, dtype=bool)
        trunc = np.zeros(self.num_envs, dtype=bool)
        out = {}
        if np.any(dones):
            final = np.empty(self.num_envs, dtype=object)
            mask = np.zeros(self.num_envs, dtype=bool)
            for i in np.nonzero(dones)[0]:
                tl = bool(infos[i].get("TimeLimit.truncated", False))
                trunc[i], term[i] = tl, not tl
                final[i] = infos[i]["terminal_observation"]
                mask[i] = True
            out["final_observation"], out["_final_observation"] = final, mask
            out["final_obs"], out["_final_obs"] = final, mask
            finfo = np.empty(self.num_envs, dtype=object)
            for i in np.nonzero(mask)[0]:
                finfo[i] = {}
            out["final_info"], out["_final_info"] = finfo, mask
        return obs, rew, term, trunc, out

    def close_extras(self, **kwargs):
        self.venv.close()

    def close(self, **kwargs):
        if self.closed:
            return
        self.close_extras(**kwargs)
        self.closed = True



# ---- gymnasium.make_vec entry points -------------------------------------------------------------------------------------------------
# gymnasium >= 1.0: ``register(id, entry_point=..., vector_entry_point=...)`` lets ``gymnasium.make_vec(id, num_envs=N, **params)``
# build a native vector environment instead of N copies of the single one (``vectorization_mode="vector_entry_point"``, the default
# when the spec has one).  pde_control_gym/__init__.py registers these factories, so the batched GPU environments are reachable
# through gymnasium's own constructor with the reference's parameter dictionary.
def _vector_factory(env_id):
    def make(num_envs: int = 1, **kwargs):
        from pde_control_gym.vector import make_vec
        for k in ("render_mode", "max_episode_steps", "disable_env_checker", "autoreset"):      # gymnasium's own keywords
            kwargs.pop(k, None)
        return GymnasiumVectorAdapter(make_vec(env_id, num_envs, **kwargs))
    make.__name__ = make.__qualname__ = "vector_" + env_id.split("-")[-1]
    make.__doc__ = f"gymnasium vector entry point of {env_id}: GymnasiumVectorAdapter(make_vec({env_id!r}, num_envs, **kwargs))."
    return make


vector_TransportPDE1D = _vector_factory("PDEControlGym-TransportPDE1D")
vector_ReactionDiffusionPDE1D = _vector_factory("PDEControlGym-ReactionDiffusionPDE1D")
vector_NavierStokes2D = _vector_factory("PDEControlGym-NavierStokes2D")
vector_TrafficPDE1D = _vector_factory("PDEControlGym-TrafficPDE1D")
vector_BurgersPDE1D = _vector_factory("PDEControlGym-BurgersPDE1D")
