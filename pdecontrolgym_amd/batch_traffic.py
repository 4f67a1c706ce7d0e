"""Batched Aw-Rascle-Zhang traffic environments (float64) on device tensors.

Mirrors the constructor arithmetic of the reference's TrafficPDE1D (environments1d/traffic_arz_env.py:24-101):
``M = len(np.arange(0, X+dx, dx))``, equilibrium velocity ``Veq(rho) = vm (1 - rho/rm)``, ``qs = rs * vs``, action
bounds ``[0.8 qs, 1.2 qs]`` fixed at construction, initial profile ``rs (sin(3 x/L pi) 0.1 + 1)``.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .checkpoint import EngineCheckpoint


class TrafficBatch(EngineCheckpoint):
    def __init__(self, T: float, dt: float, X: float, dx: float, simulation_type: str = "inlet", v_max: float = 40,
                 ro_max: float = 0.16, tau: float = 60, limit_pde_state_size: bool = False, control_freq: int = 1,
                 num_envs: int = 1, device="cuda", backend=None):
        import torch
        if simulation_type not in N.TRAFFIC_SIM:
            raise ValueError("Invalid simulation type")
        assert isinstance(control_freq, int) and control_freq >= 1, \
            f"control_freq must be a positive integer (got {control_freq} of type {type(control_freq).__name__})"
        self.T, self.dt, self.X, self.dx = T, dt, X, dx
        self.simulation_type = simulation_type
        self.vm, self.rm, self.tau = v_max, ro_max, tau
        self.limit, self.control_freq = bool(limit_pde_state_size), control_freq
        self.x = np.arange(0, X + dx, dx)
        self.M = len(self.x)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.action_dim = 2 if simulation_type == "both" else 1
        if backend is None:
            from .backend import default_backend
            backend = default_backend()
        self.backend = backend.bind(self) if hasattr(backend, "bind") else backend
        P = N.ParamsTraffic()
        P.M, P.control_freq, P.sim, P.limit = self.M, control_freq, N.TRAFFIC_SIM[simulation_type], int(self.limit)
        P.dt, P.dx, P.T, P.vm, P.rm, P.tau = dt, dx, T, v_max, ro_max, tau
        self.params = P
        B, M, dev, f64 = self.num_envs, self.M, self.device, torch.float64
        # the initial profile uses NumPy's sin so that resets are bit-identical to the reference's (:256)
        self.profile = torch.as_tensor(np.sin(3 * self.x / X * np.pi) * 0.1 + np.ones(M), dtype=f64, device=dev)
        # everything a host-facing caller reads after a step lives in ONE allocation (hostio.PackLayout): the single environment
        # fetches observation, fields, clock, reward and flags with one device-to-host copy
        from .hostio import PackLayout
        u8 = torch.uint8
        self.pack_layout = PackLayout([("obs0", (B, 2 * M), f64), ("obs1", (B, 2 * M), f64), ("r", (B, M), f64), ("y", (B, M), f64),
                                       ("time", (B,), f64), ("reward", (B,), f64), ("done", (B,), u8), ("truncated", (B,), u8)])
        self.host_pack, pv = self.pack_layout.allocate(dev)
        self.t = {
            "r": pv["r"], "y": pv["y"],
            "action": torch.zeros(B, 2, dtype=f64, device=dev), "time": pv["time"],
            "rs": torch.zeros(B, dtype=f64, device=dev), "qs_clip": torch.zeros(B, dtype=f64, device=dev),
            "obs": None, "reward": pv["reward"],
            "done": pv["done"], "truncated": pv["truncated"],
        }
        self._obs = [pv["obs0"], pv["obs1"]]
        self._flip = 0
        self.t["obs"] = self._obs[0]

    def Veq(self, rho):
        return self.vm * (1 - rho / self.rm)

    def _next_obs(self):
        self._flip ^= 1
        self.t["obs"] = self._obs[self._flip]

    def set_action_bounds(self, qs_clip):
        import torch
        self.t["qs_clip"] = torch.as_tensor(qs_clip, dtype=torch.float64, device=self.device).reshape(self.num_envs).contiguous()

    def reset(self, rs, mask=None):
        """rs [B]: steady-state density per instance; where ``mask`` is given only those instances restart."""
        import torch
        rs = torch.as_tensor(rs, dtype=torch.float64, device=self.device).reshape(self.num_envs).contiguous()
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            self.t["rs"] = torch.where(mask.bool(), rs, self.t["rs"]).contiguous()
        else:
            self.t["rs"] = rs
        self.backend.traffic_reset(self.params, self.t, self.profile, mask, self.num_envs)
        return self.t["obs"]

    def enable_auto_reset(self, rs_pool, keep_final_obs: bool = True):
        """Fused VecEnv auto-reset: an instance whose step ends done | truncated restarts inside the same launch (step and
        rollout alike) the way ``TrafficPDE1D.reset`` does, with the steady-state density of pool row (b + k*B) mod P for its
        k-th restart (``rs_pool`` [P]: the reference redraws it in 'outlet-train'); ``t['final_obs']`` keeps the last
        observation of the finished episode."""
        import torch
        B, dev = self.num_envs, self.device
        self.t["reset_rs"] = torch.as_tensor(rs_pool, dtype=torch.float64, device=dev).reshape(-1).contiguous()
        self.t["reset_profile"] = self.profile
        self.t["reset_count"] = torch.zeros(B, dtype=torch.int32, device=dev)
        self.t["final_obs"] = torch.zeros(B, 2 * self.M, dtype=torch.float64, device=dev) if keep_final_obs else None

    def disable_auto_reset(self):
        for k in ("reset_rs", "reset_profile", "reset_count", "final_obs"):
            self.t.pop(k, None)

    def can_rollout(self) -> bool:
        """``rollout`` needs the register-resident kernel (freeways of up to 64 nodes: the reference's grid has 51)."""
        return self.M <= 64 and hasattr(self.backend, "traffic_rollout")

    def policy_fits_rollout(self, policy) -> bool:
        """Whether ``policy`` (a ``FusedMLP``) can run inside the rollout kernel: 2M inputs, one output per command.  Layers of
        at most 64 units: weights + 16 observation rows within 160 KB of LDS; a layer of 65 .. 256 units: evaluated by the
        workgroup's 16 waves together on the matrix cores (bit-identical to ``FusedMLP`` itself), as in the 1D engines."""
        if not (self.can_rollout() and hasattr(policy, "layers") and hasattr(policy, "_net")):
            return False
        dims = [(int(w.shape[1]), int(w.shape[0])) for w, _, _ in policy.layers]
        D = 2 * self.M
        if dims[0][0] != D or dims[-1][1] != self.action_dim or any(o > 256 for _, o in dims):
            return False
        if any(o > 64 for _, o in dims):
            stride = lambda w: (w + 63) // 64 * 64 + 4                  # noqa: E731  (pdegym_mlp_tile.h: lds_stride)
            floats = 16 * (stride((D + 15) // 16 * 16) + 2 * stride(256)) + 32
        else:
            floats = sum((((i + 3) // 4) | 1) * 4 * o + 64 for i, o in dims) + 16 * (((D + 3) // 4) * 4 + 128)
        return 4 * floats <= 160 * 1024

    def rollout(self, obs, actions, rewards, done, truncated, policy=None, clamp="default", noise=None):
        """T env-steps in ONE launch (include/pdegym.h: pdegym_traffic_rollout): step t takes ``actions[t]`` ([T, B, action_dim]),
        writes ``obs[t + 1]`` ([T+1, B, 2M]), ``rewards[t]``, ``done[t]``, ``truncated[t]`` -- bit-identical to T ``step`` calls.
        With ``policy`` (a ``FusedMLP``) the commands are computed inside the launch from ``obs[t]`` (``obs[0]`` = the current
        observation, supplied by the caller) and written to ``actions``; ``noise`` [T, B, action_dim] float32 is added before
        the clamp.  The engine's own observation / reward / flag tensors receive the values of the last step."""
        if not self.can_rollout():
            raise ValueError("rollout needs freeways of at most 64 nodes")
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
                    raise ValueError("noise must be a contiguous float32 tensor of the actions' shape")
                net.noise, net.noise_stride = noise.data_ptr(), int(actions.shape[2])
        self.backend.traffic_rollout(self.params, self.t, obs, actions, rewards, done, truncated, self.num_envs, policy=net)
        self.t["obs"].copy_(obs[-1])
        self.t["reward"].copy_(rewards[-1])
        self.t["done"].copy_(done[-1])
        self.t["truncated"].copy_(truncated[-1])
        return obs, rewards, done, truncated

    def step(self, action):
        """action [B] / [B,1] (inlet, outlet) or [B,2] ('both'). Returns (obs [B,2M], reward, done, truncated)."""
        import torch
        from .hostio import as_kernel_input
        a = as_kernel_input(action, torch.float64, self.device, (self.num_envs, -1))     # (a pinned host tensor is read in place)
        if a.shape[1] > 2 or (self.action_dim == 2 and a.shape[1] != 2):
            raise ValueError(f"action must be [B] / [B, 1] / [B, 2] ('both' needs two columns), got {tuple(a.shape)}")
        self.t["action"] = a.contiguous()       # used in place: the kernel takes the column count as the stride
        self._next_obs()
        self.backend.traffic_step(self.params, self.t, self.num_envs)
        return self.t["obs"], self.t["reward"], self.t["done"], self.t["truncated"]
