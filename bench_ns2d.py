"""NS2D workload of bench.py: BASELINE configs[3] (NavierStokes2D 128x128, 50 Jacobi sweeps/step, batch 512, fp32)."""
from __future__ import annotations

import os
import platform
import time


class NavierStokesC4:
    name = "NavierStokes2D 128x128 K=50 B=512 fp32 (BASELINE configs[3])"
    n, B, K = 128, 512, 50
    dtype = "f32"          # "f64" = the reference's own precision (navier_stokes2D.py:186, base_env_2d.py:50)
    BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"],
          "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}

    def __init__(self, device, seed, B=None, S=None):
        import torch
        from pdecontrolgym_amd.batch2d import NSBatch2D
        self.B = B or self.B
        self.K = S or self.K
        n = self.n
        dx = 1.0 / (n - 1)
        dt = 0.2 * 0.5 * dx * dx / 0.1
        self.nt = 1000
        self.kw = dict(T=self.nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=self.BC, gamma=0.1,
                       viscosity=0.1, density=1.0, maximum_pressure_iteration=self.K)
        self.device = device
        td = self.td = torch.float64 if self.dtype == "f64" else torch.float32
        U_ref = torch.zeros(self.nt, n, n, 2, dtype=td, device=device)
        a_ref = 2.0 * torch.ones(self.nt, dtype=td, device=device)
        self.env = NSBatch2D(U_ref=U_ref, action_ref=a_ref, num_envs=self.B, device=device, dtype=td,
                             interleaved_state=os.environ.get("PDEGYM_NS_SEPARATE_UV", "0") != "1", **self.kw)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.gen = g
        c = torch.rand(self.B, 3, generator=g) * 10 - 5
        one = torch.ones(1, n, n)
        self.ic = [(c[:, k].reshape(self.B, 1, 1) * one).to(td).to(device) for k in range(3)]

    def prepare(self, total_steps):
        import torch
        self.actions = (torch.rand(total_steps, self.B, generator=self.gen) * 2 + 2).to(self.td).to(self.device)
        self.env.reset(*self.ic)
        self.i = 0

    def step(self):
        out = self.env.step(self.actions[self.i])
        self.i += 1
        return out

    def units_per_step(self):
        return self.B

    def algorithmic_bytes_per_step(self):
        return self.env.algorithmic_bytes_per_env_step() * self.B

    def compulsory_bytes_per_step(self):
        return self.env.compulsory_bytes_per_env_step() * self.B

    def config(self):
        return {"workload": self.name, "env": "PDEControlGym-NavierStokes2D", "nx": self.n, "ny": self.n,
                "batch_per_gpu": self.B, "jacobi_sweeps_per_step": self.K, "reward": "NSReward(0.1)",
                "parallelism": "independent instances, no collective"}


class NavierStokesC5(NavierStokesC4):
    """BASELINE configs[4] per-GPU shard: NavierStokes2D 256x256, 50 Jacobi sweeps/step, 512 instances per GPU, fp32."""
    name = "NavierStokes2D 256x256 K=50 B=512/GPU fp32 (BASELINE configs[4] shard)"
    n, B, K = 256, 512, 50


class NavierStokesC4F64(NavierStokesC4):
    """BASELINE configs[3] at the reference's own precision (float64 end to end, bit-exact against NumPy)."""
    name = "NavierStokes2D 128x128 K=50 B=512 fp64 (BASELINE configs[3] at the reference's precision)"
    dtype = "f64"


class NavierStokesC4B4096(NavierStokesC4):
    """BASELINE.json's metric string names "NS2D 128x128, batch 4096": configs[3] at eight times its batch on one GPU."""
    name = "NavierStokes2D 128x128 K=50 B=4096 fp32 (BASELINE metric string)"
    B = 4096


class NavierStokesC4B4096F64(NavierStokesC4B4096):
    """The metric string's "NS2D 128x128, batch 4096" at the reference's own precision (float64 end to end)."""
    name = "NavierStokes2D 128x128 K=50 B=4096 fp64 (BASELINE metric string at the reference's precision)"
    dtype = "f64"


class NavierStokesC5F64(NavierStokesC5):
    """BASELINE configs[4] per-GPU shard at the reference's own precision (float64 end to end, bit-exact against NumPy)."""
    name = "NavierStokes2D 256x256 K=50 B=512/GPU fp64 (BASELINE configs[4] shard at the reference's precision)"
    dtype = "f64"


class NavierStokesExample(NavierStokesC4):
    """The reference's own shipped NavierStokes2D configuration (examples/NavierStokes/NS2Dppo.py:36-50: 21 x 21 grid, 2000 Jacobi
    sweeps per env-step, float64) at a batch that fills the chip: one lane per grid column, three instances per wave."""
    name = "NavierStokes2D 21x21 K=2000 B=8192 fp64 (the reference's shipped example configuration)"
    n, B, K = 21, 8192, 2000
    dtype = "f64"
