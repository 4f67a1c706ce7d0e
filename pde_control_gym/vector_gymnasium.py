"""gymnasium.vector-style face over the batched environments (SURVEY.md section 8f rank 1 names both the SB3 ``VecEnv`` and the
``gymnasium.vector`` conventions).

``GymnasiumVectorAdapter(venv)`` wraps a ``PDEVecEnv`` / ``TumorVecEnv`` and speaks the gymnasium vector API:

    obs, infos = venv.reset(seed=None, options=None)
    obs, rewards, terminations, truncations, infos = venv.step(actions)

with same-step auto-reset: a finished sub-environment already returns the first observation of its next episode, and
``infos["final_observation"][i]`` holds the observation the episode ended on (``infos["_final_observation"]`` is the boolean
mask) -- the layout of gymnasium 0.29's ``SyncVectorEnv``.  Nothing is computed here; it only re-shapes what the SB3 face
returns, so the two conventions cannot drift apart.
"""
from __future__ import annotations

import numpy as np


class GymnasiumVectorAdapter:
    metadata = {"autoreset_mode": "same_step"}

    def __init__(self, venv):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.single_observation_space = venv.observation_space
        self.single_action_space = venv.action_space
        self.observation_space = venv.observation_space
        self.action_space = venv.action_space
        self.closed = False

    @property
    def unwrapped(self):
        return self.venv

    def reset(self, *, seed=None, options=None):
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
        return obs, rew, term, trunc, out

    def close(self):
        self.closed = True
        self.venv.close()
