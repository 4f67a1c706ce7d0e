"""Batched Aw-Rascle-Zhang traffic environments (float64) on device tensors.

Mirrors the constructor arithmetic of the reference's TrafficPDE1D (environments1d/traffic_arz_env.py:24-101):
``M = len(np.arange(0, X+dx, dx))``, equilibrium velocity ``Veq(rho) = vm (1 - rho/rm)``, ``qs = rs * vs``, action
bounds ``[0.8 qs, 1.2 qs]`` fixed at construction, initial profile ``rs (sin(3 x/L pi) 0.1 + 1)``.
"""
from __future__ import annotations

import numpy as np

from . import _native as N


class TrafficBatch:
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
        self.t = {
            "r": torch.zeros(B, M, dtype=f64, device=dev), "y": torch.zeros(B, M, dtype=f64, device=dev),
            "action": torch.zeros(B, 2, dtype=f64, device=dev), "time": torch.zeros(B, dtype=f64, device=dev),
            "rs": torch.zeros(B, dtype=f64, device=dev), "qs_clip": torch.zeros(B, dtype=f64, device=dev),
            "obs": None, "reward": torch.zeros(B, dtype=f64, device=dev),
            "done": torch.zeros(B, dtype=torch.uint8, device=dev), "truncated": torch.zeros(B, dtype=torch.uint8, device=dev),
        }
        self._obs = [torch.zeros(B, 2 * M, dtype=f64, device=dev) for _ in range(2)]
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

    def step(self, action):
        """action [B] / [B,1] (inlet, outlet) or [B,2] ('both'). Returns (obs [B,2M], reward, done, truncated)."""
        import torch
        a = torch.as_tensor(action, dtype=torch.float64, device=self.device).reshape(self.num_envs, -1)
        if a.shape[1] > 2 or (self.action_dim == 2 and a.shape[1] != 2):
            raise ValueError(f"action must be [B] / [B, 1] / [B, 2] ('both' needs two columns), got {tuple(a.shape)}")
        self.t["action"] = a.contiguous()       # used in place: the kernel takes the column count as the stride
        self._next_obs()
        self.backend.traffic_step(self.params, self.t, self.num_envs)
        return self.t["obs"], self.t["reward"], self.t["done"], self.t["truncated"]
