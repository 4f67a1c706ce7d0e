"""Batched 1D PDE environments (transport / reaction-diffusion) on device tensors.

``PDEBatch1D`` owns the per-instance state of B independent environments as PyTorch-ROCm tensors and
advances all of them with ONE kernel launch per env-step through the C ABI (include/pdegym.h).  It
mirrors the reference's constructor arithmetic:

    nt = int(round(T/dt)+1), nx = int(round(X/dx))          environments1d/base_env_1d.py:23-24
    S  = int(round(control_sample_rate/dt))                 environments1d/hyperbolic.py:137
    parabolic rows carry a ghost node (nx+1)                environments1d/parabolic.py:124
    sensing / control enums and their error strings         environments1d/hyperbolic.py:48-124

The full trajectory ``u[nt, nx]`` the reference keeps per environment (80 MB each at the shipped
parabolic settings) is NOT kept: only the live row plus the three scalars TunedReward1D needs
(SURVEY.md section 8a, rows R1/R2).  ``record_history=True`` restores the full history on device for
small batches (custom reward callbacks, plotting).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from . import _native as N
from .checkpoint import EngineCheckpoint


@dataclass
class RewardSpec:
    """Which reward the step kernel evaluates in its epilogue."""
    kind: int = N.REWARD_NONE
    nt: int = 0
    truncate_penalty: float = -1e-4
    terminate_reward: float = 1e2
    horizon: int = N.HORIZON_TEMPORAL      # NormReward kinds: HORIZON_DIFFERENTIAL = +||u[t] - u[t-1]||, HORIZON_T = -(mean of the
    t_horizon: int = 5                     # last t_horizon row norms) -- both evaluated by the step kernels only


_BAD_SENSING_LOC = "Invalid sensing_loc parameter. Please use 'full', 'collocated', or 'opposite'. See documentation for details."
_BAD_SENSING_TYPE = "Invalid sensing_type parameter. Please use 'Neumann' or 'Dirchilet'. See documentation for details."
_BAD_CONTROL = "Invalid control_type parameter. Please use 'Neumann' or 'Dirchilet'. See documentation for details."
_PARABOLIC_OPPOSITE_DIR = "In the parabolic PDE system, u(0, t)=0 and so Dirchilet sensing at u(0, t) is not viable. See documentation for details."


def sensing_mode(kind: str, control_type: str, sensing_loc: str, sensing_type) -> int:
    """Maps the reference's (control_type, sensing_loc, sensing_type) table onto one PDEGYM_SENSE_* code,
    raising the same bare ``Exception`` messages (hyperbolic.py:66-124, parabolic.py:66-122)."""
    if sensing_loc not in ("full", "collocated", "opposite"):
        raise Exception(_BAD_SENSING_LOC)
    if control_type not in ("Neumann", "Dirchilet"):
        raise Exception(_BAD_CONTROL)
    if sensing_loc == "full":
        return N.SENSE_FULL
    if sensing_loc == "collocated":
        return N.SENSE_LAST if control_type == "Neumann" else N.SENSE_LAST_DERIV
    if sensing_type == "Neumann":
        return N.SENSE_FIRST_DERIV
    if sensing_type == "Dirchilet":
        if kind == "parabolic":
            raise Exception(_PARABOLIC_OPPOSITE_DIR)
        return N.SENSE_FIRST
    raise Exception(_BAD_SENSING_TYPE)


class PDEBatch1D(EngineCheckpoint):
    def __init__(self, kind: str, T: float, dt: float, X: float, dx: float, control_sample_rate: float,
                 control_type: str = "Dirchilet", sensing_loc: str = "full", sensing_type="Dirchilet",
                 normalize: bool = False, max_control_value: float = 20, limit_pde_state_size: bool = False,
                 max_state_value: float = 1e10, reward: RewardSpec | None = None, num_envs: int = 1,
                 device="cuda", backend=None, record_history: bool = False, state_in_obs: bool = True, flux: str = "linear"):
        import torch
        assert kind in ("transport", "parabolic")
        if flux not in ("linear", "burgers") or (flux == "burgers" and kind != "transport"):
            raise ValueError("flux must be 'linear' (the reference's transport term) or 'burgers' (extension, transport only)")
        self.kind, self.flux = kind, flux
        self.T, self.dt, self.X, self.dx = T, dt, X, dx
        self.nt = int(round(T / dt) + 1)
        self.nx = int(round(X / dx))
        self.n = self.nx + (1 if kind == "parabolic" else 0)
        self.substeps = int(round(control_sample_rate / dt))
        self.sensing = sensing_mode(kind, control_type, sensing_loc, sensing_type)
        self.obs_dim = self.n if self.sensing == N.SENSE_FULL else 1
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.reward_spec = reward or RewardSpec()
        self.control_sample_rate, self.control_type = control_sample_rate, control_type
        self.sensing_loc, self.sensing_type, self.normalize = sensing_loc, sensing_type, bool(normalize)
        self.max_control_value, self.max_state_value = max_control_value, max_state_value
        self.limit_pde_state_size = limit_pde_state_size
        if backend is None:
            from .backend import default_backend
            backend = default_backend()
        self.backend = backend.bind(self) if hasattr(backend, "bind") else backend

        P = N.Params1D()
        P.n, P.nt, P.substeps = self.n, self.nt, self.substeps
        P.control_type = N.CONTROL[control_type]
        P.normalize = 1 if normalize else 0
        P.sensing = self.sensing
        P.limit_state = 1 if limit_pde_state_size else 0
        P.reward_kind = self.reward_spec.kind
        P.reward_nt = int(self.reward_spec.nt)
        P.reward_horizon = int(self.reward_spec.horizon) if self.reward_spec.kind >= N.REWARD_NORM_L1 else N.HORIZON_TEMPORAL
        P.reward_t_horizon = int(self.reward_spec.t_horizon) if P.reward_horizon == N.HORIZON_T else 0
        P.dt, P.dx = dt, dx                       # ctypes c_float rounds the Python double to float32
        P.F = dt / (dx ** 2)                      # parabolic.py:138, computed in double then cast
        P.rdx = 1.0 / float(C.c_float(dx).value)  # reciprocal of the float32 dx, in double (see pdegym.h)
        P.max_control = max_control_value
        P.dt64, P.dx64, P.max_control64 = dt, dx, max_control_value   # the Python doubles, for float64 operands
        P.beta_f64, P.action_kind = 0, N.ACTION_F32
        P.max_state = min(max_state_value, 3.4028234663852886e38)
        P.flux = N.FLUX_BURGERS if flux == "burgers" else N.FLUX_LINEAR
        P.truncate_penalty = self.reward_spec.truncate_penalty
        P.terminate_reward = self.reward_spec.terminate_reward
        self.params = P

        B, n, dev = self.num_envs, self.n, self.device
        f32 = torch.float32
        # reward, terminated and truncated share one allocation (4B + B + B bytes): a host-facing caller fetches the three with
        # ONE device-to-host copy (pde_control_gym.PDEVecEnv.step_wait) instead of three latency-bound ones
        self.host_pack = torch.zeros(6 * B, dtype=torch.uint8, device=dev)
        self.t = {
            "u": torch.zeros(B, n, dtype=f32, device=dev),
            "beta": torch.zeros(B, n, dtype=f32, device=dev),
            "action": torch.zeros(B, dtype=f32, device=dev),
            "time_index": torch.zeros(B, dtype=torch.int32, device=dev),
            "bsum": torch.zeros(B, dtype=torch.float64, device=dev),
            "ring": torch.zeros(B, N.RING, dtype=f32, device=dev),
            "obs": None,
            "reward": self.host_pack[:4 * B].view(f32),
            "norm_now": torch.zeros(B, dtype=f32, device=dev),
            "norm_back": torch.zeros(B, dtype=f32, device=dev),
            "terminated": self.host_pack[4 * B:5 * B],
            "truncated": self.host_pack[5 * B:],
            "history": torch.zeros(B, self.nt, n, dtype=f32, device=dev) if record_history else None,
            "reset_init": None,
            "final_obs": None,
            "reset_beta": None,
            "reset_count": None,
        }
        # observations are double-buffered: the tensor returned by step k stays valid during step k+1
        self._obs = [torch.zeros(B, self.obs_dim, dtype=f32, device=dev) for _ in range(2)]
        self._flip = 0
        self.t["obs"] = self._obs[0]
        # Full-state sensing: the observation IS the row (hyperbolic.py:72-75), so the state lives only in the observation
        # tensors -- a step reads the previous observation (bufs.state_in) and writes the next one, one row store instead of
        # two.  ``t["u"]`` then names the CURRENT observation tensor (rebound after every step / reset).
        self.state_in_obs = self.sensing == N.SENSE_FULL and not record_history and state_in_obs
        self.t["state_in"] = None
        if self.state_in_obs:
            self.t["u"] = self.t["obs"]

    # ---- state accessors ---------------------------------------------------------------------------
    @property
    def u(self):
        return self.t["u"]

    @property
    def time_index(self):
        return self.t["time_index"]

    def _next_obs(self):
        self._flip ^= 1
        self.t["obs"] = self._obs[self._flip]
        return self.t["obs"]

    # ---- API -----------------------------------------------------------------------------------------
    def set_beta(self, beta, dtype=None):
        """beta: [n] (shared) or [B, n].  Its dtype selects the arithmetic, as in the reference: float32 -> the float32
        kernels; float64 (what ``np.ones(nx)`` or an un-cast ``np.cos(...)`` gives) or an integer type -> the reference's
        mixed-precision update, evaluated in double and rounded once per stored row (hyperbolic.py:146-155,
        parabolic.py:143-144; slower kernel).  ``dtype=torch.float32`` forces the float32 path."""
        import torch
        beta = torch.as_tensor(beta)
        if dtype is None:
            dtype = torch.float32 if beta.dtype in (torch.float32, torch.float16, torch.bfloat16) else torch.float64
        beta = beta.to(device=self.device, dtype=dtype).contiguous()
        if beta.shape[-1] != self.n:
            raise ValueError(f"beta must have {self.n} nodes, got {tuple(beta.shape)}")
        self.t["beta"] = beta
        self.params.beta_f64 = 1 if dtype == torch.float64 else 0

    def reset(self, init, beta=None, mask=None):
        """(Re)start instances from ``init`` [B, n]; where ``mask`` [B] (uint8/bool) is given only those."""
        import torch
        if beta is not None:
            self.set_beta(beta)
        init = torch.as_tensor(init, dtype=torch.float32, device=self.device).contiguous()
        if init.shape != (self.num_envs, self.n):
            raise ValueError(f"init must be [{self.num_envs}, {self.n}], got {tuple(init.shape)}")
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            # masked reset writes into the CURRENT obs buffer so untouched instances keep their observation
        else:
            self._next_obs()
        if self.state_in_obs:
            self.t["u"] = None                      # the reset rows go to the observation buffer only
        self.backend.reset1d(self.params, self.t, init, mask, self.num_envs)
        if self.state_in_obs:
            self.t["u"] = self.t["obs"]
        return self.t["obs"]

    def enable_auto_reset(self, init_pool, keep_final_obs: bool = True, beta_pool=None):
        """Fused VecEnv auto-reset: an instance that ends a step terminated|truncated restarts inside the same kernel launch
        (no host round trip).  ``init_pool`` is a caller-owned [P, n] tensor with P >= B rows; the k-th restart of instance b
        takes row (b + k*B) mod P, so with P > B consecutive episodes of one instance start from different rows.  With
        ``beta_pool`` [P, n] (same dtype as the current beta) the plant parameter is redrawn from the same row -- the
        reference calls BOTH reset callbacks at every reset (hyperbolic.py:207-209).  Pools may be refreshed between steps
        (e.g. re-sampled on device)."""
        import torch
        pool = torch.as_tensor(init_pool, dtype=torch.float32, device=self.device).contiguous()
        if pool.dim() != 2 or pool.shape[1] != self.n or pool.shape[0] < self.num_envs:
            raise ValueError(f"init_pool must be [P >= {self.num_envs}, {self.n}], got {tuple(pool.shape)}")
        self.t["reset_init"] = pool
        self.t["reset_count"] = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.t["reset_beta"] = None
        if beta_pool is not None:
            if self.t["beta"].dim() == 1:        # a shared row cannot be redrawn per instance: give every instance its own
                self.t["beta"] = self.t["beta"].unsqueeze(0).repeat(self.num_envs, 1).contiguous()
            bp = torch.as_tensor(beta_pool).to(device=self.device, dtype=self.t["beta"].dtype).contiguous()
            if bp.shape != pool.shape:
                raise ValueError(f"beta_pool must have the shape of init_pool {tuple(pool.shape)}, got {tuple(bp.shape)}")
            self.t["reset_beta"] = bp
        self.t["final_obs"] = (torch.zeros(self.num_envs, self.obs_dim, dtype=torch.float32, device=self.device)
                               if keep_final_obs else None)

    def disable_auto_reset(self):
        self.t["reset_init"] = None
        self.t["final_obs"] = None
        self.t["reset_beta"] = None
        self.t["reset_count"] = None

    def step(self, action, out_obs=None, out_reward=None, out_terminated=None, out_truncated=None, action_kind=None):
        """Advance every instance by one env-step (S sub-steps). action: [B] tensor.
        Returns (obs, reward, terminated, truncated) device tensors (uint8 flags).  The ``out_*`` tensors, when given,
        receive the outputs directly (contiguous, right dtype/shape) -- e.g. slot t of a rollout buffer.
        ``action_kind``: how NumPy would type the reference's ``control`` argument (_native.ACTION_F32 / _F64 / _WEAK, see
        include/pdegym.h); default: float64 tensors -> ACTION_F64, everything else -> float32."""
        import torch
        action = torch.as_tensor(action)
        if action_kind is None:
            action_kind = N.ACTION_F64 if action.dtype == torch.float64 else N.ACTION_F32
        adt = torch.float32 if action_kind == N.ACTION_F32 else torch.float64
        if action.dtype == adt and action.device == self.device and action.dim() == 1 and action.shape[0] == self.num_envs and action.is_contiguous():
            a = action                      # already what the kernel reads (the VecEnv face's staging tensor): no torch dispatch
        else:
            a = action.to(device=self.device, dtype=adt).reshape(self.num_envs).contiguous()
        self.t["action"] = a
        self.params.action_kind = action_kind
        prev = self.t["obs"]
        if out_obs is not None:
            self.t["obs"] = out_obs.view(self.num_envs, self.obs_dim)
        else:
            self._next_obs()
            if self.state_in_obs and self.t["obs"] is prev:   # never write the observation over the state it is computed from
                self._next_obs()
        if self.state_in_obs:
            if self.t["obs"].data_ptr() == prev.data_ptr():
                raise ValueError("out_obs must not be the tensor that holds the current observation (it is the state)")
            self.t["state_in"], self.t["u"] = prev, None
        for key, out in (("reward", out_reward), ("terminated", out_terminated), ("truncated", out_truncated)):
            if out is not None:
                self.t[key] = out
        self.backend.step1d(self.kind, self.params, self.t, self.num_envs)
        if self.state_in_obs:
            self.t["u"] = self.t["obs"]
        return self.t["obs"], self.t["reward"], self.t["terminated"], self.t["truncated"]

    # ---- batch-of-one face: command in / results out through ONE pinned host allocation ------------------------------
    def enable_host_io(self):
        """Host-facing mode of a small batch (the single environments, ``num_envs=1``): the command and everything a host caller
        reads after a step -- observation, reward, ||u||, terminated, truncated -- live in ONE pinned host allocation that is
        mapped into the device's address space (hipHostMalloc).  The step kernel reads the command from it and writes its
        results into it directly, so an env-step is ONE stream operation (the kernel launch) + ONE stream synchronisation: no
        host-to-device copy of the command, no device-to-host copies of the results, no torch dispatch.  The plant state (row,
        beta, ring, running sums, time index, history) stays in HBM.  Not for ``state_in_obs`` engines (their observation IS
        the state).  Returns the NumPy views ``{"obs", "reward", "norm_now", "terminated", "truncated"}`` (valid after
        ``step_host`` / ``sync_host``; overwritten by the next step)."""
        import torch
        if self.state_in_obs:
            raise ValueError("host I/O needs the state in its own tensor (build the engine with state_in_obs=False)")
        if self.num_envs != 1:
            # (the command slot is 8 bytes wide so that one address serves float32 and float64 commands: an array of float32
            # commands would need 4-byte slots -- batches take the staged path of PDEVecEnv)
            raise ValueError("host I/O is the batch-of-one face (num_envs = 1)")
        if getattr(self, "_hio", None) is not None:
            return self._hio["np"]
        B, od = self.num_envs, self.obs_dim
        sizes = (("action", 8 * B), ("obs", 4 * B * od), ("reward", 4 * B), ("norm_now", 4 * B), ("terminated", B), ("truncated", B))
        pack = torch.zeros((sum(s for _, s in sizes) + 63) // 64 * 64, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        raw, off, sl = pack.numpy(), 0, {}
        for k, nb in sizes:
            sl[k] = slice(off, off + nb)
            off += nb
        f32 = torch.float32
        tv = {"action": pack[sl["action"]].view(torch.float64), "obs": pack[sl["obs"]].view(f32).view(B, od),
              "reward": pack[sl["reward"]].view(f32), "norm_now": pack[sl["norm_now"]].view(f32),
              "terminated": pack[sl["terminated"]], "truncated": pack[sl["truncated"]]}
        import numpy as np
        nv = {"obs": raw[sl["obs"]].view(np.float32).reshape(B, od), "reward": raw[sl["reward"]].view(np.float32),
              "norm_now": raw[sl["norm_now"]].view(np.float32), "terminated": raw[sl["terminated"]],
              "truncated": raw[sl["truncated"]]}
        # one 8-byte slot per instance: a float32 command occupies its first four bytes, a float64 one all eight (the kernel reads
        # the slot as params.action_kind says)
        a64 = raw[sl["action"]].view(np.float64)
        a32 = raw[sl["action"]].view(np.float32)[::2]
        for k in ("obs", "reward", "norm_now", "terminated", "truncated"):
            self.t[k] = tv[k]
        self._obs = [tv["obs"], tv["obs"]]
        self._hio = {"pack": pack, "action": tv["action"], "a32": a32, "a64": a64, "np": nv, "call": None, "key": None}
        return nv

    def _host_call(self):
        """The prepared C-ABI call of ``step_host`` (argument structures filled once; rebuilt when a tensor of the dictionary
        has been replaced, e.g. a new beta at reset)."""
        io = self._hio
        self.t["action"] = io["action"]
        key = (self.t["beta"].data_ptr(), self.t["u"].data_ptr(), None if self.t["history"] is None else self.t["history"].data_ptr(),
               int(self.params.beta_f64))
        if io["call"] is None or io["key"] != key:
            prep = getattr(self.backend, "prepare_step1d", None)
            if prep is not None:
                io["call"] = prep(self.kind, self.params, self.t, self.num_envs)
            else:                        # (the CPU tests' oracle double: the plain entry point, the command as a typed tensor)
                def call():
                    import torch
                    f32 = self.params.action_kind == N.ACTION_F32
                    self.t["action"] = torch.from_numpy((io["a32"] if f32 else io["a64"]).copy())
                    self.backend.step1d(self.kind, self.params, self.t, self.num_envs)
                io["call"] = call
            io["key"] = key
        return io["call"]

    def sync_host(self):
        """Wait until the results of the last launch are in the host views."""
        if self.device.type == "cuda":
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    def step_host(self, value: float, action_kind: int = N.ACTION_F32):
        """One env-step of a batch of one commanded from the host: ``value`` (a Python float) is written into the pinned command
        slot in the precision ``action_kind`` names, the step kernel is launched, the stream is synchronised.  The results are in
        the views ``enable_host_io`` returned."""
        io = self._hio
        if action_kind == N.ACTION_F32:
            io["a32"][:] = value
        else:
            io["a64"][:] = value
        self.params.action_kind = action_kind
        self._host_call()()
        self.sync_host()

    def can_rollout(self) -> bool:
        """True when ``rollout`` applies: every control / sensing combination (round 4: Neumann actuation and scalar sensing
        too), as long as the state has ONE home -- the observation slots with full-state sensing (``state_in_obs``), ``u``
        with scalar sensing -- no history is recorded, the operands are float32, the rows are register-resident, and the
        reward is one the rollout kernels evaluate (not the "differential" horizon)."""
        return bool((self.state_in_obs or self.sensing != N.SENSE_FULL) and self.t["history"] is None and not self.params.beta_f64
                    and self.params.reward_horizon == N.HORIZON_TEMPORAL
                    and self.n <= N.MAX_N1D_REG and hasattr(self.backend, "rollout1d"))

    def policy_fits_rollout(self, policy) -> bool:
        """Whether ``policy`` (a ``FusedMLP``) can be evaluated inside the rollout kernel: the observation (the row of at most
        513 nodes, or the one sensed value) as its input and one output.  Layers of at most 64 units: one neuron per lane, the
        weights + 16 observation rows within 160 KB of LDS.  A layer of 65..256 units (SB3's 256-256 actors): the 16 waves of a
        workgroup evaluate the network together on the matrix cores, weights streamed from L2 -- bit-identical to
        ``pdegym_mlp_forward`` (16 observation rows + the hidden rows within 160 KB of LDS: any row the kernel takes)."""
        if not (self.can_rollout() and hasattr(policy, "layers") and hasattr(policy, "_net")):
            return False
        dims = [(int(w.shape[1]), int(w.shape[0])) for w, _, _ in policy.layers]
        if dims[0][0] != self.obs_dim or self.n > 513 or dims[-1][1] != 1 or any(o > 256 for _, o in dims):
            return False
        stride = lambda w: (w + 63) // 64 * 64 + 4                  # noqa: E731  (pdegym_mlp_tile.h: lds_stride)
        if any(o > 64 for _, o in dims):
            floats = 16 * (stride((self.obs_dim + 15) // 16 * 16) + 2 * stride(256)) + 32
        else:
            floats = sum((((i + 3) // 4) | 1) * 4 * o + 64 for i, o in dims) + 16 * (((self.obs_dim + 3) // 4) * 4 + 128)
        return 4 * floats <= 160 * 1024

    def rollout(self, obs, actions, rewards, terminated, truncated, policy=None, clamp="default", noise=None, obs_noise=None,
                obs_seen=None):
        """T env-steps in ONE launch (include/pdegym.h: pdegym_*_rollout): step t takes the commands from ``actions[t]`` and
        writes ``obs[t + 1]`` ([T+1, B, obs_dim]), ``rewards[t]``, ``terminated[t]``, ``truncated[t]`` -- bit-identical to T
        calls of ``step(actions[t], out_obs=obs[t + 1], ...)``, fused auto-reset included.  With full-state sensing ``obs[0]``
        is the state the rollout starts from and the engine's current observation (its state) is a copy of ``obs[T]``
        afterwards; with scalar sensing the state is the engine's own ``u`` (advanced in place) and ``obs[0]`` only feeds a
        policy.

        ``policy`` (a ``FusedMLP`` with layers of at most 256 units and one output, see ``policy_fits_rollout``): evaluated
        inside the launch on ``obs[t]`` (+ ``obs_noise[t]`` [T, B, obs_dim], the pre-drawn sensing noise; ``obs_seen[t]``
        receives what the policy read); ``actions[t]`` then RECEIVES the command (after ``noise[t]`` [T, B] float32 and the
        clamp).  A policy with a layer of 65 .. 256 units is evaluated by the 16 waves of a workgroup in ``policy.forward_into``'s
        own MFMA reduction order: commands -- hence whole trajectories -- are bit-identical to the two-launch loop
        (``forward_into`` + ``step``).  A policy whose layers all have <= 64 units sums each neuron in one fmaf chain instead:
        its commands agree with ``forward_into`` to float32 rounding (rtol ~2e-5), NOT bit for bit, and trajectories drift apart
        accordingly -- the environment arithmetic is bit-identical to step calls either way (given the same commands)."""
        if not self.can_rollout():
            raise ValueError("rollout needs a state with one home (full-state sensing: state_in_obs; no history) and float32 operands")
        self.params.action_kind = N.ACTION_F32
        net = None
        if policy is not None:
            if not self.policy_fits_rollout(policy):
                raise ValueError("this policy cannot run inside the rollout kernel (see policy_fits_rollout)")
            import torch
            if not (obs.is_cuda and torch.cuda.is_current_stream_capturing()):
                policy.refresh()             # pick up in-place parameter updates (as FusedMLP.forward_into does)
            net = policy._net(policy.clamp if clamp == "default" else clamp)
            if noise is not None:
                import torch
                if noise.dtype != torch.float32 or tuple(noise.shape) != tuple(actions.shape) or not noise.is_contiguous():
                    raise ValueError("noise must be a contiguous float32 [T, B] tensor")
                net.noise, net.noise_stride = noise.data_ptr(), 1
        elif obs_noise is not None or obs_seen is not None:
            raise ValueError("obs_noise / obs_seen shape the policy's input: they need a policy")
        self.backend.rollout1d(self.kind, self.params, self.t, obs, actions, rewards, terminated, truncated, self.num_envs,
                               policy=net, obs_noise=obs_noise, obs_seen=obs_seen)
        self.t["obs"].copy_(obs[-1])
        if self.state_in_obs:
            self.t["u"] = self.t["obs"]
        return obs, rewards, terminated, truncated

    # ---- checkpoint / resume (pdecontrolgym_amd/checkpoint.py) ---------------------------------------------
    def _checkpoint_meta(self):
        # the reward kind / horizon decide what the norm ring HOLDS (TunedReward1D: 2-norms of look-back rows; NormReward "t-horizon":
        # the reward's own norm of every recent row), sensing / control what the observation and the boundary sums mean: a
        # checkpoint of one configuration must not load silently into another
        return {"engine": "PDEBatch1D", "kind": self.kind, "flux": self.flux, "num_envs": self.num_envs, "n": self.n,
                "nt": self.nt, "substeps": self.substeps, "obs_dim": self.obs_dim, "state_in_obs": bool(self.state_in_obs),
                "reward_kind": int(self.params.reward_kind), "reward_horizon": int(self.params.reward_horizon),
                "reward_t_horizon": int(self.params.reward_t_horizon), "sensing": int(self.sensing),
                "control_type": str(self.control_type)}

    def _after_load(self, sd):
        import torch
        self.params.beta_f64 = 1 if self.t["beta"].dtype == torch.float64 else 0
        if self.state_in_obs:            # the row lives in the observation tensor: "u" names it, it is not a second tensor
            self.t["u"] = self.t["obs"]

    # ---- roofline bookkeeping ---------------------------------------------
    def algorithmic_bytes_per_env_step(self) -> int:
        """Streaming model: each sub-step reads the previous row and beta and writes the new row
        (12 B per node), plus the observation row and ~16 B of scalars per env-step."""
        return self.substeps * 12 * self.n + 4 * self.n + 16

    def compulsory_bytes_per_env_step(self) -> int:
        """What the fused kernel must move: row in, beta in, row out, obs out (+ scalars); with the state in the observation
        tensors the row is written once."""
        return 4 * self.n * (2 if self.state_in_obs else 3) + 4 * self.obs_dim + 64
