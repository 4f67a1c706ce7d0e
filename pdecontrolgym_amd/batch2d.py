"""Batched 2D Navier-Stokes environments on device tensors.

``NSBatch2D`` owns u, v, p of B independent instances ([B, ny, nx], row = y, col = x) and advances all of
them with one kernel launch per env-step through the C ABI.  Constructor arithmetic mirrors the reference:

    nt = int(round(T/dt)), nx = int(round(X/dx+1)), ny = int(round(Y/dy+1))    environments2d/base_env_2d.py:27-29
    RuntimeError("Stability is not guarenteed") if dt > stable_factor*0.5*min(dx,dy)**2/nu   navier_stokes2D.py:56-58
    boundary_condition = {"upper"|"lower"|"left"|"right": [cond_u, cond_v]}     navier_stokes2D.py:61-91

dtype=float64 is the parity mode (the reference is float64 end to end); dtype=float32 is the throughput
mode asked for by BASELINE.json (tolerance stated in tests/test_gpu_ns2d.py).
"""
from __future__ import annotations

from . import _native as N
from .checkpoint import EngineCheckpoint


def _bc_codes(boundary_condition: dict):
    codes = []
    for edge in N.EDGES:
        pair = boundary_condition[edge]
        row = []
        for c in pair:
            if c not in N.BC:
                # the reference's `match` silently ignores unknown strings (navier_stokes2D.py:79-90); a typo there
                # leaves the edge untouched, which no kernel can reproduce meaningfully -> fail loudly instead
                raise Exception(f"Invalid boundary condition {c!r}. Please use 'Neumann', 'Dirchilet' or 'Controllable'.")
            row.append(N.BC[c])
        codes.append(row)
    return codes


class NSBatch2D(EngineCheckpoint):
    def __init__(self, T: float, dt: float, X: float, dx: float, Y: float, dy: float, boundary_condition: dict,
                 U_ref, action_ref, action_dim: int = 1, gamma: float = 0.1, viscosity: float = 0.1,
                 density: float = 1.0, maximum_pressure_iteration: int = 2000, stable_factor: float = 0.5,
                 num_envs: int = 1, device="cuda", dtype=None, backend=None, interleaved_state: bool = True):
        import torch
        self.nt = int(round(T / dt))
        self.nx = int(round(X / dx + 1))
        self.ny = int(round(Y / dy + 1))
        self.dt, self.dx, self.dy = dt, dx, dy
        max_t = (0.5 * min(dx, dy) ** 2 / viscosity)
        if dt > stable_factor * max_t:
            raise RuntimeError("Stability is not guarenteed")
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.dtype = dtype or torch.float64
        self.action_dim = int(action_dim)
        self.iters = int(maximum_pressure_iteration)
        self.ctor = dict(T=T, dt=dt, X=X, dx=dx, Y=Y, dy=dy, boundary_condition=boundary_condition, gamma=gamma,
                         viscosity=viscosity, density=density, maximum_pressure_iteration=self.iters,
                         stable_factor=stable_factor)
        if backend is None:
            from .backend import default_backend
            backend = default_backend()
        self.backend = backend.bind(self) if hasattr(backend, "bind") else backend

        P = N.ParamsNS2D()
        P.nx, P.ny, P.nt, P.iters, P.action_dim = self.nx, self.ny, self.nt, self.iters, self.action_dim
        codes = _bc_codes(boundary_condition)
        for e in range(4):
            for k in range(2):
                P.bc[e][k] = codes[e][k]
        P.dt, P.dx, P.dy = dt, dx, dy
        P.viscosity, P.density, P.gamma = viscosity, density, gamma
        self.params = P

        B, ny, nx, dev, dt_ = self.num_envs, self.ny, self.nx, self.device, self.dtype
        U_ref = torch.as_tensor(U_ref, dtype=dt_, device=dev).contiguous()
        if U_ref.dim() != 4 or U_ref.shape[1:] != (ny, nx, 2):
            raise ValueError(f"U_ref must be [nt, {ny}, {nx}, 2], got {tuple(U_ref.shape)}")
        a_ref = torch.as_tensor(action_ref, dtype=dt_, device=dev).reshape(-1).contiguous()
        # interleaved_state: the velocity state lives only in the (double-buffered) observation tensors
        # [B, ny, nx, 2] -- like the reference, whose obs U[t] IS the state -- which saves writing u and v separately
        self.interleaved_state = bool(interleaved_state)
        self.t = {
            "u": None if self.interleaved_state else torch.zeros(B, ny, nx, dtype=dt_, device=dev),
            "v": None if self.interleaved_state else torch.zeros(B, ny, nx, dtype=dt_, device=dev),
            "state_in": None,
            "p": torch.zeros(B, ny, nx, dtype=dt_, device=dev),
            "p_out": None,
            "reset_u0": None, "reset_v0": None, "reset_p0": None, "final_obs": None, "reset_count": None,
            "scratch": torch.zeros(B, 4, ny, nx, dtype=dt_, device=dev),
            "action": torch.zeros(B, self.action_dim, dtype=dt_, device=dev),
            "time_index": torch.zeros(B, dtype=torch.int32, device=dev),
            "U_ref": U_ref,
            "action_ref": a_ref,
            "obs": None,
            "reward": torch.zeros(B, dtype=dt_, device=dev),
            "terminated": torch.zeros(B, dtype=torch.uint8, device=dev),
        }
        self._obs = [torch.zeros(B, ny, nx, 2, dtype=dt_, device=dev) for _ in range(2)]
        self._flip = 0
        self.t["obs"] = self._obs[0]
        # the 256x256 pipeline finishes its pressure solve in a second buffer: ping-pong two pressure tensors instead of
        # copying the result home every step (include/pdegym.h: p_out)
        # (float64: the last slab pass writes its own rows of the solved pressure into a field it does not read, include/pdegym.h)
        self._p_pingpong = self.interleaved_state and nx == 256 and ny == 256
        if self._p_pingpong:
            self.t["p_out"] = torch.zeros(B, ny, nx, dtype=dt_, device=dev)

    @property
    def u(self):
        """Current u field [B, ny, nx] (a strided view of the observation tensor in interleaved mode)."""
        return self.t["obs"][..., 0] if self.interleaved_state else self.t["u"]

    @property
    def v(self):
        return self.t["obs"][..., 1] if self.interleaved_state else self.t["v"]

    @property
    def p(self):
        return self.t["p"]

    @property
    def time_index(self):
        return self.t["time_index"]

    def _next_obs(self, out_obs=None):
        prev = self.t["obs"]
        if self.interleaved_state:
            self.t["state_in"] = prev               # the observation just produced is the next call's input state
        if out_obs is not None:                     # the caller's buffer (e.g. slot t+1 of a rollout) receives the observation
            self.t["obs"] = out_obs.view(self.num_envs, self.ny, self.nx, 2)
            return
        self._flip ^= 1
        if self._obs[self._flip] is prev:           # never write the observation over the state it is computed from
            self._flip ^= 1
        self.t["obs"] = self._obs[self._flip]

    def enable_auto_reset(self, u0_pool, v0_pool, p0_pool, keep_final_obs: bool = True):
        """Fused VecEnv auto-reset: an instance whose step ends terminated restarts inside the same C-ABI call (no host
        round trip) from row (b + k*B) mod P of the pools ([P >= B, ny, nx] each; k = restarts of instance b so far).  The
        observation returned for it is the first one of the new episode; the last one of the old episode is kept in
        ``t["final_obs"]`` (SB3's ``terminal_observation``)."""
        import torch
        pools = [torch.as_tensor(a, dtype=self.dtype, device=self.device).contiguous() for a in (u0_pool, v0_pool, p0_pool)]
        for a in pools:
            if a.dim() != 3 or a.shape[0] < self.num_envs or tuple(a.shape[1:]) != (self.ny, self.nx) or a.shape != pools[0].shape:
                raise ValueError(f"pools must be [P >= {self.num_envs}, {self.ny}, {self.nx}], got {tuple(a.shape)}")
        self.t["reset_u0"], self.t["reset_v0"], self.t["reset_p0"] = pools
        self.t["reset_count"] = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.t["final_obs"] = (torch.zeros(self.num_envs, self.ny, self.nx, 2, dtype=self.dtype, device=self.device)
                               if keep_final_obs else None)

    def disable_auto_reset(self):
        for k in ("reset_u0", "reset_v0", "reset_p0", "final_obs", "reset_count"):
            self.t[k] = None

    def reset(self, u0, v0, p0, mask=None):
        import torch
        cvt = lambda a: torch.as_tensor(a, dtype=self.dtype, device=self.device).expand(self.num_envs, self.ny, self.nx).contiguous()
        u0, v0, p0 = cvt(u0), cvt(v0), cvt(p0)
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        # a reset writes the (new) state into the CURRENT observation buffer: untouched instances keep theirs
        self.backend.ns2d_reset(self.params, self.t, u0, v0, p0, mask, self.num_envs)
        return self.t["obs"]

    def step(self, action, out_obs=None, out_reward=None, out_terminated=None):
        """action: [B] or [B, action_dim]. Returns (obs [B,ny,nx,2], reward [B], terminated [B] uint8).  The ``out_*``
        tensors, when given, receive the outputs directly (e.g. slot t of a rollout buffer)."""
        import torch
        a = torch.as_tensor(action, dtype=self.dtype, device=self.device).reshape(self.num_envs, self.action_dim).contiguous()
        self.t["action"] = a
        self._next_obs(out_obs)
        if out_reward is not None:
            self.t["reward"] = out_reward
        if out_terminated is not None:
            self.t["terminated"] = out_terminated
        self.backend.ns2d_step(self.params, self.t, self.num_envs)
        if self._p_pingpong:        # the solved pressure is in p_out: it becomes p (the warm start of the next step)
            self.t["p"], self.t["p_out"] = self.t["p_out"], self.t["p"]
        return self.t["obs"], self.t["reward"], self.t["terminated"]

    # ---- batch-of-one face: command in / results out through ONE pinned host allocation ------------------------------
    def enable_host_io(self):
        """Host-facing mode of a small batch (the single environment): the command, the observation, the reward and the terminated
        flag live in ONE pinned host allocation mapped into the device's address space -- the kernels read the command from it and
        write their results into it, so an env-step is the step's launches + ONE stream synchronisation, no copies in either
        direction.  Needs the velocity state in its own tensors (``interleaved_state=False``: the observation is a pure output).
        Returns the NumPy views ``{"action", "obs", "reward", "terminated"}``."""
        import numpy as np
        import torch
        if self.interleaved_state:
            raise ValueError("host I/O needs the velocity state in its own tensors (interleaved_state=False)")
        if getattr(self, "_hio", None) is not None:
            return self._hio["np"]
        B, ny, nx, A = self.num_envs, self.ny, self.nx, self.action_dim
        w = 8 if self.dtype == torch.float64 else 4
        sizes = (("action", w * B * A), ("obs", w * B * ny * nx * 2), ("reward", w * B), ("terminated", B))
        pack = torch.zeros((sum((nb + 7) // 8 * 8 for _, nb in sizes) + 63) // 64 * 64, dtype=torch.uint8,
                           pin_memory=self.device.type == "cuda")
        raw, off, tv, nv = pack.numpy(), 0, {}, {}
        npdt = np.float64 if w == 8 else np.float32
        shapes = {"action": (B, A), "obs": (B, ny, nx, 2), "reward": (B,)}
        for k, nb in sizes:
            if k == "terminated":
                tv[k], nv[k] = pack[off:off + nb], raw[off:off + nb]
            else:
                tv[k] = pack[off:off + nb].view(self.dtype).view(shapes[k])
                nv[k] = raw[off:off + nb].view(npdt).reshape(shapes[k])
            off += (nb + 7) // 8 * 8
        for k in ("action", "obs", "reward", "terminated"):
            self.t[k] = tv[k]
        self._obs = [tv["obs"], tv["obs"]]
        self._hio = {"pack": pack, "np": nv, "call": None}
        return nv

    def sync_host(self):
        if self.device.type == "cuda":
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    def step_host(self):
        """One env-step commanded from the host: the caller has written the command into the ``"action"`` view; launches the step
        and synchronises the stream.  Results are in the views ``enable_host_io`` returned."""
        io = self._hio
        if io["call"] is None:
            prep = getattr(self.backend, "prepare_ns2d_step", None)
            io["call"] = (prep(self.params, self.t, self.num_envs) if prep is not None
                          else (lambda: self.backend.ns2d_step(self.params, self.t, self.num_envs)))
        io["call"]()
        self.sync_host()

    def can_rollout(self) -> bool:
        """True when ``rollout`` applies: the column-per-lane kernel's grids (8, 11, 16, 21, 26, 31 or 32 rows, at most 64
        columns -- the reference's shipped 21 x 21 example among them) with the state in the observation tensors."""
        return bool(self.interleaved_state and self.ny in (8, 11, 16, 21, 26, 31, 32) and 3 <= self.nx <= 64
                    and hasattr(self.backend, "ns2d_rollout"))

    def rollout(self, obs, actions, rewards, terminated):
        """T env-steps in ONE launch (include/pdegym.h: pdegym_ns2d_rollout_*), commands given ahead: step t takes ``actions[t]``
        ([T, B, action_dim]) and writes ``obs[t + 1]`` ([T+1, B, ny, nx, 2]; ``obs[0]`` = the state the rollout starts from),
        ``rewards[t]``, ``terminated[t]`` -- fields, flags and counters bit-identical to T calls of ``step(actions[t],
        out_obs=obs[t + 1], ...)``, fused auto-reset included.  Rewards are bit-identical to step calls that run on the same
        (column-per-lane) kernel: float32 always, float64 from 400 instances per 1024 SIMDs; smaller float64 batches step on the
        workgroup kernel, which sums the reward in another order (equal to ~1e-15 relative; include/pdegym.h).  Afterwards the
        engine's current observation (its state) is a copy of ``obs[T]``."""
        if not self.can_rollout():
            raise ValueError("rollout needs one of the column kernel's grids (8 / 11 / 16 / 21 / 26 / 31 / 32 rows, <= 64 columns) "
                             "and the interleaved state layout")
        pingpong = self._p_pingpong
        self.backend.ns2d_rollout(self.params, self.t, obs, actions, rewards, terminated, self.num_envs)
        assert not pingpong                      # (only the 256 x 256 grids ping-pong their pressure; they cannot roll out)
        self.t["obs"].copy_(obs[-1])
        self.t["reward"].copy_(rewards[-1])
        self.t["terminated"].copy_(terminated[-1])
        return obs, rewards, terminated

    def solve_pressure(self, u, v, p_prev):
        """K Jacobi sweeps of the pressure Poisson problem for arbitrary fields (navier_stokes2D.py:94-116).
        u, v, p_prev: [M, ny, nx] device tensors; returns a new [M, ny, nx] tensor."""
        import torch
        u = torch.as_tensor(u, dtype=self.dtype, device=self.device).contiguous()
        v = torch.as_tensor(v, dtype=self.dtype, device=self.device).contiguous()
        p_prev = torch.as_tensor(p_prev, dtype=self.dtype, device=self.device).contiguous()
        M = u.shape[0]
        out = torch.empty_like(p_prev)
        scratch = torch.empty(M, 2, self.ny, self.nx, dtype=self.dtype, device=self.device)
        self.backend.ns2d_solve_pressure(self.params, u, v, p_prev, out, scratch, M)
        return out

    # ---- checkpoint / resume (pdecontrolgym_amd/checkpoint.py) ---------------------------------------------
    def _checkpoint_meta(self):
        return {"engine": "NSBatch2D", "num_envs": self.num_envs, "nx": self.nx, "ny": self.ny, "nt": self.nt,
                "dtype": str(self.dtype), "interleaved_state": self.interleaved_state}

    # ---- roofline bookkeeping ---------------------------------------------
    def algorithmic_bytes_per_env_step(self) -> int:
        word = 4 if str(self.dtype).endswith("float32") else 8
        return word * (3 * self.iters + 16) * self.nx * self.ny

    def compulsory_bytes_per_env_step(self) -> int:
        """What a fused step must move: interleaved state (the observation of the previous step IS the state) 2 fields + p in,
        observation 2 + p out = 6 fields; with separate u, v the state is read and written next to the observation: 10 fields.
        The shared reference frame comes from L2."""
        word = 4 if str(self.dtype).endswith("float32") else 8
        return word * (6 if self.interleaved_state else 10) * self.nx * self.ny
