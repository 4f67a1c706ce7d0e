"""Batched 1D brain-tumour radiotherapy environments (float64) on device tensors.

Mirrors the constructor arithmetic of the reference's BrainTumor1D (environments1d/brain_tumor_env.py:33-104):
``nx = int(round(X/dx)+1)``, ``xScale = np.linspace(0, X, nx)``, detection thresholds ``ratio * k``.  One
``step()`` is one simulated day for every patient in the batch (one kernel launch).
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .checkpoint import EngineCheckpoint


class TumorBatch(EngineCheckpoint):
    MARGIN = 25     # mm added to the T2 radius for the treated region (brain_tumor_env.py:257)

    def __init__(self, T: float, dt: float, X: float, dx: float, total_dosage: float,
                 t1_detection_threshold: float = 0.8, t2_detection_threshold: float = 0.16,
                 dosage_termination_threshold: float = 0.1, D: float = 0.2, rho: float = 0.03, alpha: float = 0.04,
                 alpha_beta_ratio: float = 10, k: float = 1e5, t1_detection_radius: float = 15,
                 t1_death_radius: float = 35, num_envs: int = 1, device="cuda", backend=None, record_history: bool = False):
        import torch
        self.T, self.dt, self.X, self.dx = T, dt, X, dx
        self.nt = int(round(T / dt) + 1)
        self.nx = int(round(X / dx) + 1)
        self.xScale = np.linspace(0, X, self.nx)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.total_dosage = float(total_dosage)
        self.alpha, self.alpha_beta_ratio, self.k = alpha, alpha_beta_ratio, k
        if backend is None:
            from .backend import default_backend
            backend = default_backend()
        self.backend = backend.bind(self) if hasattr(backend, "bind") else backend
        P = N.ParamsTumor()
        P.nx, P.nt = self.nx, self.nt
        P.dt, P.dx, P.dx2 = dt, dx, dx ** 2
        P.D, P.rho, P.alpha, P.alpha_beta_ratio, P.k = D, rho, alpha, alpha_beta_ratio, k
        P.thr_t1, P.thr_t2 = t1_detection_threshold * k, t2_detection_threshold * k
        P.detect_radius, P.death_radius = t1_detection_radius, t1_death_radius
        P.total_dosage, P.dose_end, P.margin = self.total_dosage, dosage_termination_threshold, self.MARGIN
        self.params = P
        B, dev, f64, i32 = self.num_envs, self.device, torch.float64, torch.int32
        # everything a host-facing caller reads after a day lives in ONE allocation (hostio.PackLayout): one device-to-host copy
        from .hostio import PackLayout
        u8 = torch.uint8
        self.pack_layout = PackLayout([("u", (B, self.nx), f64), ("out", (B, 4), f64), ("remaining", (B,), f64), ("reward", (B,), f64),
                                       ("time_index", (B,), i32), ("stage", (B,), i32), ("days", (B, 5), i32),
                                       ("terminated", (B,), u8), ("truncated", (B,), u8)])
        self.host_pack, pv = self.pack_layout.allocate(dev)
        pv["remaining"].fill_(self.total_dosage)
        self.t = {
            "u": pv["u"],
            "xscale": torch.as_tensor(self.xScale, dtype=f64, device=dev),
            "control": torch.zeros(B, dtype=f64, device=dev), "kill": None,
            "time_index": pv["time_index"], "stage": pv["stage"],
            "remaining": pv["remaining"],
            "days": pv["days"],
            "t_benchmark": torch.full((B,), float("nan"), dtype=f64, device=dev),
            "reward": pv["reward"],
            "terminated": pv["terminated"],
            "truncated": pv["truncated"],
            "out": pv["out"],
            "active": None,
            # optional trajectory on the device: history[b, t] = row of day t, t1_log[b, t] = T1 radius / dx (NaN = invisible)
            "history": torch.zeros(B, self.nt, self.nx, dtype=f64, device=dev) if record_history else None,
            "t1_log": torch.full((B, self.nt), float("nan"), dtype=f64, device=dev) if record_history else None,
        }
        self.t["days"][:, 4] = -1

    def set_benchmark(self, t_benchmark):
        """Baseline survival days per patient (NaN = not set: every reward is 0, brain_tumor_reward.py:43-47)."""
        import torch
        if torch.is_tensor(t_benchmark) and t_benchmark.dtype == torch.float64 and t_benchmark.numel() == self.num_envs and \
                t_benchmark.device.type == "cpu" and self.device.type == "cuda" and t_benchmark.is_pinned():
            self.t["t_benchmark"] = t_benchmark.reshape(self.num_envs)          # pinned host memory: read in place by the kernel
            return
        tb = torch.as_tensor(t_benchmark, dtype=torch.float64, device=self.device)
        self.t["t_benchmark"] = tb.expand(self.num_envs).contiguous() if tb.dim() == 0 else tb.reshape(self.num_envs).contiguous()

    def reset(self, init, mask=None):
        """init [nx] (shared) or [B, nx]; where ``mask`` is given only those patients restart."""
        import torch
        init = torch.as_tensor(init, dtype=torch.float64, device=self.device).contiguous()
        if init.dim() == 2:
            assert init.shape == (self.num_envs, self.nx)
        else:
            assert init.shape == (self.nx,)
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self.backend.tumor_reset(self.params, self.t, init, mask, self.num_envs)
        if self.t["history"] is not None:
            sel = slice(None) if mask is None else mask.bool()
            self.t["history"][sel] = 0
            self.t["history"][sel, 0] = self.t["u"][sel]
            self.t["t1_log"][sel] = float("nan")
        return self.t["u"]

    def _set_active(self, active):
        import torch
        self.t["active"] = None if active is None else \
            torch.as_tensor(active, device=self.device).to(torch.uint8).reshape(self.num_envs).contiguous()

    def advance(self, mode: int, max_days: int = None, active=None):
        """Run whole stretches of days inside ONE launch (control 0 every day): ``N.TUMOR_RUN_GROWTH`` -- patients in Growth
        until their stage changes; ``N.TUMOR_RUN_POST`` -- patients in Post-Therapy until death or the time limit;
        ``N.TUMOR_RUN_TO_END`` -- everybody until death or the time limit (the open-loop benchmark).  Outputs
        (reward, flags, ``out``) are those of each patient's last simulated day; others are left untouched."""
        self._set_active(active)
        self.backend.tumor_advance(self.params, self.t, int(mode), int(self.nt if max_days is None else max_days), self.num_envs)
        return self.t["u"], self.t["reward"], self.t["terminated"], self.t["truncated"]

    def step(self, control, kill=None, active=None):
        """control [B]: proportion of total_dosage requested today.  ``kill`` [B] optionally carries
        ``1 - exp(-alpha*BED)`` evaluated by the caller (NumPy bit parity); by default the kernel evaluates it.
        Returns (u [B,nx] -- the live state, updated in place --, reward, terminated, truncated)."""
        import torch
        from .hostio import as_kernel_input
        self.t["control"] = as_kernel_input(control, torch.float64, self.device, (self.num_envs,))   # (pinned host tensors: in place)
        self.t["kill"] = None if kill is None else as_kernel_input(kill, torch.float64, self.device, (self.num_envs,))
        self._set_active(active)
        self.backend.tumor_step(self.params, self.t, self.num_envs)
        return self.t["u"], self.t["reward"], self.t["terminated"], self.t["truncated"]
