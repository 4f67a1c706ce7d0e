"""CPU oracle, un-batched: ONE environment stepped the way the reference steps it (TEST INFRASTRUCTURE ONLY).

``oracle/pde_oracle.py`` restates the reference's arithmetic batched over a leading instance axis, which costs it masks and
``np.where`` selections the reference does not pay.  This module restates the same arithmetic for a single environment with the
reference's own cost profile -- the full trajectory ``u[nt, n]`` kept in memory, one Python ``while`` iteration and one handful of
slice ufuncs per PDE sub-step -- so that ``bench.py``'s ``single_env`` block can time "what ``env.step()`` costs in NumPy" beside the
batch-of-one GPU face on the same shape ("missing" 1).  Only ``tests/`` and ``bench.py``'s CPU legs import it.

Pinned by ``tests/test_oracle_golden.py``: bit-identical to the batched oracle (itself pinned by the reference-generated golden
vectors) and directly to the golden vectors H1 / P1.

Restated from (reference checkout):
  environments1d/base_env_1d.py:23-24,36-39   nt / nx arithmetic, the ``normalize`` map
  environments1d/hyperbolic.py:126-169        transport step: boundary node first, then the upwind update of nodes 0 .. nx-2
  environments1d/parabolic.py:126-164         reaction-diffusion step on nx+1 nodes, u(0,t) = 0, boundary node last
  environments1d/hyperbolic.py:171-194        terminate / truncate
  rewards/tuned_reward_1d.py:17-40            the three-branch reward on the kept history
"""
from __future__ import annotations

import numpy as np

LOOKBACK = 100     # tuned_reward_1d.py:25,40


class SingleEnv1D:
    """kind "transport" | "parabolic"; full-state sensing; Dirichlet or Neumann actuation; TunedReward1D (or no reward)."""

    def __init__(self, kind, T, dt, X, dx, control_sample_rate, control_type="Dirchilet", normalize=False, max_control_value=20,
                 limit_pde_state_size=False, max_state_value=1e10, reward=None):
        assert kind in ("transport", "parabolic") and control_type in ("Dirchilet", "Neumann")
        self.kind, self.dt, self.dx = kind, dt, dx
        self.nt = int(round(T / dt) + 1)
        self.nx = int(round(X / dx))
        self.n = self.nx + (kind == "parabolic")
        self.S = int(round(control_sample_rate / dt))
        self.neumann = control_type == "Neumann"
        self.scale = max_control_value if normalize else None
        self.limit, self.max_state = limit_pde_state_size, max_state_value
        self.reward = reward                                     # (nt, truncate_penalty, terminate_reward) or None

    def reset(self, init, beta):
        self.u = np.zeros((self.nt, self.n), dtype=np.float32)
        self.u[0] = init
        self.beta = beta
        self.t = 0
        return self.u[0]

    def _boundary(self, control, neighbour):
        c = control * self.dx + neighbour if self.neumann else control
        return (c + 1) * self.scale - self.scale if self.scale is not None else c

    def step(self, control):
        u, nx, dt, dx, beta = self.u, self.nx, self.dt, self.dx, self.beta
        done = 0
        if self.kind == "transport":
            while done < self.S and self.t < self.nt - 1:
                self.t += 1
                new, old = u[self.t], u[self.t - 1]
                new[-1] = self._boundary(control, new[-2])       # (the NEW row's neighbour: still zero, hyperbolic.py:143-145)
                new[0:nx - 1] = old[0:nx - 1] + dt * ((old[1:nx] - old[0:nx - 1]) / dx + (old[0] * beta)[0:nx - 1])
                done += 1
        else:
            F = dt / (dx ** 2)
            while done < self.S and self.t < self.nt - 1:
                self.t += 1
                new, old = u[self.t], u[self.t - 1]
                new[1:nx] = old[1:nx] + F * (old[0:nx - 1] - 2 * old[1:nx] + old[2:nx + 1]) + dt * beta[1:nx] * old[1:nx]
                new[0] = 0
                new[-1] = self._boundary(control, old[-2])
                done += 1
        row = u[self.t]
        terminate = self.t >= self.nt - 1
        truncate = bool(self.limit and np.linalg.norm(row, 2) >= self.max_state)
        rew = None
        if self.reward is not None:
            r_nt, pen, bonus = self.reward
            if terminate and np.linalg.norm(row) < 20:
                rew = bonus - np.sum(abs(u[:, -1])) / 1000 - np.linalg.norm(row)
            elif truncate:
                rew = pen * (r_nt - self.t)
            else:
                rew = np.linalg.norm(u[self.t - LOOKBACK]) - np.linalg.norm(row)
        return row, rew, terminate, truncate
