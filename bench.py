#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched PDE-environment stepper on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (one rank per GPU; instances are independent, so there is NO data-path collective: the only
  communication is the barrier / max-over-ranks of the timing).

A "step" = one env-step of EVERY instance of the batch = one launch of the fused step kernel
(S PDE sub-steps + norms + reward + observation + auto-reset), inputs already resident in HBM.
Default workload = BASELINE.json configs[1]: ReactionDiffusionPDE1D ("Parabolic1D") nx=256, batch 4096
per GPU, fp32, S=100 sub-steps per env-step.  Other workloads (--workload NAME prints NAME's own line;
the default run measures all of them too: a compact `secondary` list in the printed line, everything in bench_also.json): transport_c3, burgers_c3 (extension), parabolic_c2_policy_loop (C2 with its MLP
controller evaluated on the device every step), parabolic_c2_rollout (the same loop as ONE kernel per 25 env-steps: here a
"step" is one launch and `value` still counts env-steps), parabolic_c2_policy_loop_256 / parabolic_c2_rollout_256 (the same two with
the 257-256-256-1 ReLU actor SB3's SAC builds), parabolic_c2_open_loop_rollout (25 env-steps per launch, commands given ahead),
parabolic_c2_s1 / parabolic_c2_s1_open_loop_rollout / parabolic_c2_s1_rollout (SURVEY 8d "and also S = 1": one sub-step per env-step,
per-step launch and 100 env-steps per launch without / with the policy inside), ns2d_c4, ns2d_c4_f64, ns2d_c4_b4096,
ns2d_c4_f64_b4096, ns2d_c5, ns2d_c5_f64 (the _f64 lines: the same workloads at the reference's own precision), ns2d_example (the
reference's shipped 21x21 K=2000 float64 configuration), traffic_arz, traffic_arz_rollout (25 env-steps per launch), brain_tumor;
bench_also.json additionally carries vecenv_host: the SB3-facing PDEVecEnv.step (NumPy in / out, PCIe-inclusive) at the C2 shape,
single_env: microseconds per env.step() of ONE environment through the drop-in face beside the un-batched NumPy oracle (bench_single.py),
and hbm_probe: the measured copy / read / fill rate of the box (tools/hbm_probe.hip).

Prints ONE JSON line (rank 0), kept under 4 KB (final_line): metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline + secondary (+ the per-rank fields for N > 1).
The complete result of the run -- every workload's configuration, region times and full roofline block -- is written to
bench_also.json next to this script (and to gpurun_out/ when that directory exists).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def _dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("PDEGYM_BENCH_SHARE_GPU") == "1":      # test hook: several ranks on one GPU (exercises the N>1 code path)
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The step path has no collective; the process group only carries the timing barrier and one MAX all-reduce.
        # RCCL ("nccl") first, gloo as a fallback so a fabric hiccup cannot lose the measurement.
        try:
            if os.environ.get("PDEGYM_BENCH_FAIL_NCCL") == "1":      # test hook: the fallback below is otherwise unreachable on a healthy box
                raise RuntimeError("PDEGYM_BENCH_FAIL_NCCL=1: simulated RCCL initialisation failure")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
        except Exception as ex:
            sys.stderr.write(f"rank {rank}: RCCL process group failed ({ex!r}); using gloo for the timing barrier\n")
            try:
                dist.destroy_process_group()
            except Exception:
                pass
            dist.init_process_group(backend="gloo")
    return rank, local, world


def _barrier(world):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
class Parabolic1D:
    """BASELINE config 2: ReactionDiffusionPDE1D nx=256 (257 nodes), F=dt/dx^2=0.25, S=100, B=4096/GPU."""
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=100 (BASELINE configs[1])"
    kind, nx, B, S, amp, glo, ghi = "parabolic", 256, 4096, 100, 50.0, 7.5, 8.5
    dtype = "f32"
    flux, max_control, ic_lo, ic_hi, act_lo = "linear", 20, 1.0, 10.0, -1.0

    def __init__(self, device, seed, B=None, S=None):
        import torch
        from pdecontrolgym_amd import _native as N
        from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
        self.B = B or self.B
        self.S = S or self.S
        nx = self.nx
        dx = 1.0 / nx
        dt = 0.25 * dx * dx if self.kind == "parabolic" else 0.5 * dx
        self.kw = dict(T=1000 * self.S * dt, dt=dt, X=1, dx=dx, control_sample_rate=self.S * dt, control_type="Dirchilet",
                       sensing_loc="full", sensing_type=None, normalize=True, max_control_value=self.max_control,
                       limit_pde_state_size=True, max_state_value=1e10)
        nt1 = int(round(self.kw["T"] / dt))
        self.reward_args = (nt1, -1e3, 3e2)
        self.env = PDEBatch1D(self.kind, reward=RewardSpec(N.REWARD_TUNED1D, *self.reward_args), num_envs=self.B,
                              device=device, flux=self.flux, **self.kw)
        n = self.env.n
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.linspace(0, 1, n, dtype=torch.float64)
        gam = torch.rand(self.B, 1, generator=g, dtype=torch.float64) * (self.ghi - self.glo) + self.glo
        self.beta = (self.amp * torch.cos(gam * torch.acos(x))).float().to(device)
        c = torch.rand(self.B, 1, generator=g) * (self.ic_hi - self.ic_lo) + self.ic_lo
        self.init = (c * torch.ones(1, n)).float().to(device)
        c2 = torch.rand(self.B, 1, generator=g) * (self.ic_hi - self.ic_lo) + self.ic_lo
        self.pool = (c2 * torch.ones(1, n)).float().to(device)
        self.gen = g
        self.device = device

    def prepare(self, total_steps):
        import torch
        self.actions = (torch.rand(total_steps, self.B, generator=self.gen) * (1 - self.act_lo) + self.act_lo).float().to(self.device)
        self.env.reset(self.init, self.beta)
        self.env.enable_auto_reset(self.pool, keep_final_obs=True)
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
        return {"workload": self.name, "env": "PDEControlGym-ReactionDiffusionPDE1D" if self.kind == "parabolic" else
                ("PDEControlGym-BurgersPDE1D (extension)" if self.flux == "burgers" else "PDEControlGym-TransportPDE1D"),
                "nx": self.nx, "nodes": self.env.n, "batch_per_gpu": self.B, "substeps_per_env_step": self.S,
                "reward": "TunedReward1D", "auto_reset": "fused", "parallelism": "independent instances, no collective"}


class ParabolicPolicyLoop(Parabolic1D):
    """the C2 environment with its controller in the loop -- an SB3-style MlpPolicy (257-64-64-1, tanh;
    transport1Dppo.py:88-90) evaluated on the observation of the previous step by pdegym_mlp_forward (one launch: forward pass,
    action clamp, store), then the env-step with fused auto-reset: two launches per env-step, nothing on the host."""
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=100 with a 257-64-64-1 tanh MLP policy in the loop (FusedMLP + env-step)"
    HIDDEN, ACT = 64, "tanh"

    def prepare(self, total_steps):
        import torch
        from pdecontrolgym_amd.policy import FusedMLP
        super().prepare(total_steps)
        torch.manual_seed(0)
        n, H = self.env.n, self.HIDDEN
        act = torch.nn.Tanh if self.ACT == "tanh" else torch.nn.ReLU
        net = torch.nn.Sequential(torch.nn.Linear(n, H), act(), torch.nn.Linear(H, H), act(),
                                  torch.nn.Linear(H, 1), torch.nn.Tanh()).to(self.device)
        self.policy = FusedMLP(net, clamp=(-1.0, 1.0))
        self.act = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        self.obs = self.env.t["obs"]

    def step(self):
        self.policy.forward_into(self.obs, self.act)
        out = self.env.step(self.act)
        self.obs = out[0]
        return out

    def config(self):
        c = super().config()
        c["policy"] = f"MLP 257-{self.HIDDEN}-{self.HIDDEN}-1 {self.ACT}, float32, pdegym_mlp_forward (MFMA), clamp to [-1, 1] fused"
        return c


class ParabolicRollout(ParabolicPolicyLoop):
    """The same policy-in-the-loop rollout as ONE kernel per CHUNK env-steps (pdegym_parabolic_rollout with the policy inside,
    include/pdegym.h): a bench "step" is one launch = CHUNK env-steps of every instance; `value` counts env-steps."""
    CHUNK = 25
    name = ("ReactionDiffusionPDE1D nx=256 B=4096 S=100, 257-64-64-1 tanh MLP policy, 25 env-steps per launch "
            "(pdegym_parabolic_rollout: policy + env-step + auto-reset inside one kernel)")

    def prepare(self, total_steps):
        import torch
        super().prepare(1)
        T, B, n = self.CHUNK, self.B, self.env.n
        self.robs = torch.zeros(T + 1, B, n, device=self.device)
        self.robs[0].copy_(self.env.t["obs"])
        self.ract = torch.zeros(T, B, device=self.device)
        self.rrew = torch.zeros(T, B, device=self.device)
        self.rte = torch.zeros(T, B, dtype=torch.uint8, device=self.device)
        self.rtr = torch.zeros(T, B, dtype=torch.uint8, device=self.device)
        assert self.env.policy_fits_rollout(self.policy)

    def step(self):
        self.env.rollout(self.robs, self.ract, self.rrew, self.rte, self.rtr, policy=self.policy)
        self.robs[0].copy_(self.robs[self.CHUNK])         # the next chunk starts where this one ended
        self.i += 1

    def units_per_step(self):
        return self.B * self.CHUNK

    def algorithmic_bytes_per_step(self):
        return super().algorithmic_bytes_per_step() * self.CHUNK

    def compulsory_bytes_per_step(self):
        return super().compulsory_bytes_per_step() * self.CHUNK

    def config(self):
        c = super().config()
        c["policy"] = (f"MLP 257-{self.HIDDEN}-{self.HIDDEN}-1 {self.ACT}, float32, evaluated inside the rollout kernel "
                       + ("(weights in LDS, one fma chain per neuron)" if self.HIDDEN <= 64 else
                          "(16 waves per workgroup together: pdegym_mlp_forward's MFMA reduction, weights streamed from L2)"))
        c["env_steps_per_launch"] = self.CHUNK
        return c


class ParabolicPolicyLoop256(ParabolicPolicyLoop):
    """The same loop with the actor SB3's SAC builds (two hidden layers of 256 ReLU units, tanh-squashed output;
    reactionDiffusion1Dsac.py:95): pdegym_mlp_forward with 16 waves per workgroup + the env-step."""
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=100 with a 257-256-256-1 ReLU MLP policy in the loop (FusedMLP + env-step)"
    HIDDEN, ACT = 256, "relu"


class ParabolicRollout256(ParabolicRollout):
    """... and as ONE kernel per 25 env-steps: the 16 waves of a workgroup evaluate the 256-256 actor together inside the rollout
    kernel (pdegym_policy.h: eval_wide), bit-identical to the two-launch loop above."""
    name = ("ReactionDiffusionPDE1D nx=256 B=4096 S=100, 257-256-256-1 ReLU MLP policy, 25 env-steps per launch "
            "(pdegym_parabolic_rollout: policy + env-step + auto-reset inside one kernel)")
    HIDDEN, ACT = 256, "relu"


class ParabolicOpenLoopRollout(Parabolic1D):
    """C2 with the commands of CHUNK env-steps handed over at once (an open-loop controller, or actions sampled ahead):
    pdegym_parabolic_rollout without a policy, one launch per CHUNK env-steps, bit-identical to CHUNK step launches."""
    CHUNK = 25
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=100, 25 env-steps per launch (pdegym_parabolic_rollout, commands given ahead)"

    def prepare(self, total_steps):
        import torch
        super().prepare(total_steps * self.CHUNK)
        T, B, n = self.CHUNK, self.B, self.env.n
        self.robs = torch.zeros(T + 1, B, n, device=self.device)
        self.robs[0].copy_(self.env.t["obs"])
        self.rrew = torch.zeros(T, B, device=self.device)
        self.rte = torch.zeros(T, B, dtype=torch.uint8, device=self.device)
        self.rtr = torch.zeros(T, B, dtype=torch.uint8, device=self.device)

    def step(self):
        T = self.CHUNK
        self.env.rollout(self.robs, self.actions[self.i * T:(self.i + 1) * T], self.rrew, self.rte, self.rtr)
        self.robs[0].copy_(self.robs[T])
        self.i += 1

    def units_per_step(self):
        return self.B * self.CHUNK

    def algorithmic_bytes_per_step(self):
        return super().algorithmic_bytes_per_step() * self.CHUNK

    def compulsory_bytes_per_step(self):
        return super().compulsory_bytes_per_step() * self.CHUNK

    def config(self):
        c = super().config()
        c["env_steps_per_launch"] = self.CHUNK
        return c


class ParabolicS1(Parabolic1D):
    """"S = 100 and also S = 1" -- the C2 environment with ONE PDE sub-step per env-step (control_sample_rate =
    dt): 4 128 algorithmic bytes per env-step, the launch-bound regime (one launch per env-step costs more than its arithmetic)."""
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=1 (SURVEY 8d: the launch-bound regime), one launch per env-step"
    S = 1


class ParabolicS1OpenLoopRollout(ParabolicOpenLoopRollout):
    """S = 1 with the commands of 100 env-steps handed over at once: one launch per 100 env-steps (pdegym_parabolic_rollout)."""
    S, CHUNK = 1, 100
    name = "ReactionDiffusionPDE1D nx=256 B=4096 S=1, 100 env-steps per launch (pdegym_parabolic_rollout, commands given ahead)"


class ParabolicS1Rollout(ParabolicRollout):
    """S = 1 with the MLP controller inside the rollout kernel: policy + env-step + auto-reset, 100 env-steps per launch -- the
    configuration SURVEY section 8f rank 1 says the whole-node figure is "actually won" in."""
    S, CHUNK = 1, 100
    name = ("ReactionDiffusionPDE1D nx=256 B=4096 S=1, 257-64-64-1 tanh MLP policy, 100 env-steps per launch "
            "(pdegym_parabolic_rollout: policy + env-step + auto-reset inside one kernel)")


class Transport1D(Parabolic1D):
    """BASELINE config 3 shape: TransportPDE1D nx=512, dt=0.5dx, S=100, B=16384/GPU (the reference has no
    Burgers env)."""
    name = "TransportPDE1D nx=512 B=16384 S=100 (BASELINE configs[2] shape)"
    kind, nx, B, S, amp, glo, ghi = "transport", 512, 16384, 100, 5.0, 7.0, 7.7


class Burgers1D(Transport1D):
    """EXTENSION (not in the reference, parity unpinned): BASELINE config 3 as worded -- nonlinear flux u u_x on the
    transport kernel, nx=512, S=100, B=16384; amplitudes keep dt*max|u|/dx <= 1."""
    name = "BurgersPDE1D (extension, not in the reference) nx=512 B=16384 S=100 (BASELINE configs[2] wording)"
    flux, max_control, ic_lo, ic_hi, amp, act_lo = "burgers", 1.0, 0.1, 1.0, 0.0, 0.1     # u stays in (0, 1]: monotone upwind


# ------------------------------------------------------------------------------------------------
# CPU baseline: the NumPy oracle run like the reference (ONE instance, Python loop over sub-steps).  NumPy only --
# it runs before this process touches the GPU, and the all-cores figure uses plain child processes.
# ------------------------------------------------------------------------------------------------
def cpu_port_rate(workload_key, seconds, seed=0):
    import numpy as np
    from oracle import pde_oracle as po
    rng = np.random.default_rng(seed)
    if workload_key.startswith("parabolic_c2"):      # the policy-in-the-loop variants: the CPU port times the environment alone
        workload_key = "parabolic_c2_s1" if workload_key.startswith("parabolic_c2_s1") else "parabolic_c2"
    if workload_key == "traffic_arz":
        env = po.TrafficOracle(240, 0.25, 500, 10, "outlet", 40, 0.16, 60, True, TrafficARZ.S)
        rs = [0.12]
        qs = 0.12 * 40 * (1 - 0.12 / 0.16)
        acts = rng.uniform(0.8, 1.2, (256, 1, 1)) * qs
        reset = lambda: env.reset(rs)
        done = lambda out: bool(out[2][0] or out[3][0] or env.time_index[0] >= 239)
        what = f"float64, {TrafficARZ.S} sub-steps each"
    elif workload_key == "brain_tumor":
        env = po.BrainTumorOracle(600, 1, 200, 1, 61.2)
        xs = np.linspace(0, 200, 201)
        ic = (0.8 * 1e5 * np.exp(-0.25 * (xs ** 2)))[None]
        acts = rng.uniform(0, 0.05, (256, 1))
        reset = lambda: env.reset(ic, [363.0])
        done = lambda out: bool(out[2][0] or out[3][0])
        what = "float64, one simulated day each"
    elif workload_key in ("parabolic_c2", "parabolic_c2_s1", "transport_c3", "burgers_c3"):
        cls = {"parabolic_c2": Parabolic1D, "parabolic_c2_s1": ParabolicS1, "transport_c3": Transport1D, "burgers_c3": Burgers1D}[workload_key]
        nx, S = cls.nx, cls.S
        dx = 1.0 / nx
        dt = 0.25 * dx * dx if cls.kind == "parabolic" else 0.5 * dx
        okw = dict(T=1000 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full",
                   sensing_type=None, normalize=True, max_control_value=cls.max_control, limit_pde_state_size=True, max_state_value=1e10)
        ocl = po.ParabolicOracle if cls.kind == "parabolic" else (po.BurgersOracle if cls.flux == "burgers" else po.TransportOracle)
        env = ocl(reward=po.TunedReward1DOracle(int(round(okw["T"] / dt)), -1e3, 3e2), keep_history=False, **okw)
        n = nx + (cls.kind == "parabolic")
        init = (rng.uniform(cls.ic_lo, cls.ic_hi) * np.ones((1, n))).astype(np.float32)
        beta = (cls.amp * np.cos(rng.uniform(cls.glo, cls.ghi) * np.arccos(np.linspace(0, 1, n))))[None].astype(np.float32)
        acts = rng.uniform(cls.act_lo, 1, (256, 1)).astype(np.float32)
        reset = lambda: env.reset(init, beta)
        done = lambda out: bool(out[2][0] or out[3][0])
        what = f"{S} sub-steps each"
    else:
        from bench_ns2d import NavierStokesC4, NavierStokesC5
        W = NavierStokesC5 if workload_key in ("ns2d_c5", "ns2d_c5_f64") else NavierStokesC4
        nn, K, nt = W.n, W.K, 1000
        dx = 1.0 / (nn - 1)
        dt = 0.2 * 0.5 * dx * dx / 0.1
        env = po.NavierStokesOracle(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=W.BC, U_ref=np.zeros((nt, nn, nn, 2)),
                                    action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K)
        ic = [rng.uniform(-5, 5) * np.ones((1, nn, nn)) for _ in range(3)]
        acts = rng.uniform(2, 4, (256, 1))
        reset = lambda: env.reset(*ic)
        done = lambda out: bool(env.time_index[0] >= nt - 2)
        what = f"float64, {K} Jacobi sweeps each"
    reset()
    with np.errstate(all="ignore"):
        for i in range(10):                      # warm-up (also skips the CPU denormal slow start of NS)
            env.step(acts[i])
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            out = env.step(acts[k % len(acts)])
            k += 1
            if done(out):
                reset()
    el = time.perf_counter() - t0
    return k / el, k, el, what


def cpu_baseline_report(workload_key, seconds):
    """Single-process figure (what a user of the reference gets) + all-cores figure (P independent worker processes)."""
    import subprocess
    rate, k, el, what = cpu_port_rate(workload_key, seconds)
    rep = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{k} env-steps of ONE instance ({what}) in a Python loop over the NumPy oracle, {el:.1f} s on "
                     f"{platform.processor() or platform.machine()}; B instances on one core run at the same aggregate rate"}
    try:
        P = len(os.sched_getaffinity(0))
        sec = min(seconds, 8.0)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", workload_key, "--cpu-seconds", str(sec),
                                   "--seed", str(100 + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
                 for i in range(P)]
        tot = 0.0
        for pr in procs:
            o, _ = pr.communicate(timeout=sec * 4 + 120)
            tot += float(o.decode().strip().splitlines()[-1])
        rep["all_cores"] = {"value": tot, "logical_cpus": P,
                            "sample": f"{P} independent worker processes x {sec:.0f} s, one instance each, one per LOGICAL cpu of the "
                                      "affinity mask (SMT siblings and a shared memory system: heavily contended -- the aggregate is "
                                      f"{tot / max(rate, 1e-9):.1f}x the single-core figure, not {P}x)"}
    except Exception as ex:
        rep["all_cores"] = {"error": repr(ex)}
    return rep


WORKLOADS = {"parabolic_c2": Parabolic1D, "transport_c3": Transport1D, "burgers_c3": Burgers1D}
class TrafficARZ:
    """TrafficPDE1D (reference notebook configuration T=240, dt=0.25, dx=10, X=500, M=51 nodes,
    float64), 'outlet' control, control_freq=2 sub-steps per env-step, B=16384 instances."""
    name = "TrafficPDE1D M=51 f64 B=16384 control_freq=2 (reference notebook configuration)"
    dtype = "f64"
    B, S = 16384, 2

    def __init__(self, device, seed, B=None, S=None):
        import torch
        from pdecontrolgym_amd.batch_traffic import TrafficBatch
        self.B, self.S = B or self.B, S or self.S
        self.device = device
        self.env = TrafficBatch(240, 0.25, 500, 10, "outlet", 40, 0.16, 60, True, self.S, num_envs=self.B, device=device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.gen = g
        self.rs = torch.tensor([0.115, 0.12, 0.125], dtype=torch.float64)[torch.randint(0, 3, (self.B,), generator=g)]
        self.qs = self.rs * (40 * (1 - self.rs / 0.16))
        self.env.set_action_bounds(self.qs)

    def prepare(self, total_steps):
        import torch
        a = (torch.rand(total_steps, self.B, 1, generator=self.gen, dtype=torch.float64) * 0.4 + 0.8) * self.qs.reshape(1, -1, 1)
        self.actions = a.to(self.device).contiguous()          # [T, B, 1]: one outlet command per freeway, used in place
        self.rs_dev = self.rs.to(self.device)
        self.env.reset(self.rs_dev)
        self.i = 0

    def begin_region(self):
        """Start of every timed region (inside the captured graph): fresh episodes.  An episode lasts T / dt = 960 simulated
        seconds = 3840 env-steps and a finished freeway is not advanced any more, so a graph replayed past that point (the
        untimed warm replays alone are > 2000 env-steps) would time launches that do nothing."""
        self.env.reset(self.rs_dev)        # (a device tensor: nothing crosses PCIe inside the captured graph)
        self.since_reset = 0

    # The reference advances `time` by dt per env-step and runs the sub-steps only `while self.time < self.T` (traffic_arz_env.py:146,
    # :172) while the episode ends at time >= T / dt (:106, seconds compared with a step count -- kept): with T = 240 s, dt = 0.25 s the
    # PDE is advanced during the first 960 env-steps and merely re-observed during the following 2880.  A region must stay inside the
    # live part, or it times launches whose sub-steps are skipped (round 5: 310 us per rollout launch of 25 live env-steps, 170 us beyond).
    LIVE_ENV_STEPS = 950

    def _stay_live(self, env_steps):
        """Restart the freeways (inside the captured graph) before `env_steps` more would leave the live part of the episode."""
        if getattr(self, "since_reset", 0) + env_steps > self.LIVE_ENV_STEPS:
            self.env.reset(self.rs_dev)
            self.since_reset = 0
        self.since_reset = getattr(self, "since_reset", 0) + env_steps

    def step(self):
        self._stay_live(1)
        out = self.env.step(self.actions[self.i])
        self.i += 1
        return out

    def units_per_step(self):
        return self.B

    def algorithmic_bytes_per_step(self):      # streaming model: r, y read + written per sub-step (f64), obs (r, v), scalars
        M = self.env.M
        return (self.S * 32 * M + 16 * M + 48) * self.B

    def compulsory_bytes_per_step(self):
        M = self.env.M
        return (2 * 16 * M + 16 * M + 64) * self.B

    def config(self):
        return {"workload": self.name, "env": "PDEControlGym-TrafficPDE1D", "nodes": self.env.M, "batch_per_gpu": self.B,
                "substeps_per_env_step": self.S, "reward": "TrafficARZReward", "parallelism": "independent instances, no collective"}


class TrafficARZRollout(TrafficARZ):
    """The same freeways with the commands of CHUNK env-steps handed over at once: pdegym_traffic_rollout, one launch per CHUNK
    env-steps with (r, y) in registers across them, bit-identical to CHUNK step launches; a bench "step" is one launch."""
    CHUNK = 25
    name = "TrafficPDE1D M=51 f64 B=16384 control_freq=2, 25 env-steps per launch (pdegym_traffic_rollout, commands given ahead)"

    def prepare(self, total_steps):
        import torch
        super().prepare(total_steps * self.CHUNK)
        T, B, M = self.CHUNK, self.B, self.env.M
        f64 = torch.float64
        self.robs = torch.zeros(T + 1, B, 2 * M, dtype=f64, device=self.device)
        self.rrew = torch.zeros(T, B, dtype=f64, device=self.device)
        self.rdn = torch.zeros(T, B, dtype=torch.uint8, device=self.device)
        self.rtr = torch.zeros(T, B, dtype=torch.uint8, device=self.device)

    def step(self):
        T = self.CHUNK
        self._stay_live(T)       # long regions (--steps 200 = 5000 env-steps): restart before the live part of the episode ends
        self.env.rollout(self.robs, self.actions[self.i * T:(self.i + 1) * T], self.rrew, self.rdn, self.rtr)
        self.i += 1

    def units_per_step(self):
        return self.B * self.CHUNK

    def algorithmic_bytes_per_step(self):
        return super().algorithmic_bytes_per_step() * self.CHUNK

    def compulsory_bytes_per_step(self):
        return super().compulsory_bytes_per_step() * self.CHUNK

    def config(self):
        c = super().config()
        c["env_steps_per_launch"] = self.CHUNK
        return c


class BrainTumor:
    """BrainTumor1D (reference notebook configuration T=600, X=200, dt=dx=1, nx=201, float64,
    total_dosage=61.2), one simulated day per env-step, B=65536 patients with individual daily doses."""
    name = "BrainTumor1D nx=201 f64 B=65536 (reference notebook configuration)"
    dtype = "f64"
    B, S = 65536, 1

    def __init__(self, device, seed, B=None, S=None):
        import numpy as np
        import torch
        from pdecontrolgym_amd.batch_tumor import TumorBatch
        self.B = B or self.B
        self.device = device
        self.env = TumorBatch(600, 1, 200, 1, 61.2, num_envs=self.B, device=device)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        xs = np.linspace(0, 200, 201)
        self.init = torch.as_tensor(0.8 * 1e5 * np.exp(-0.25 * (xs ** 2)), dtype=torch.float64, device=device)
        self.env.set_benchmark(363.0)

    COHORT_DAYS = 350      # untreated patients die around day 363 (t_benchmark): a cohort is replaced before anyone has

    def prepare(self, total_steps):
        import torch
        self.actions = (torch.rand(min(total_steps, 64), self.B, generator=self.gen, dtype=torch.float64) * 0.05).to(self.device)
        self.env.reset(self.init)
        self.i = self.day = 0

    def begin_region(self):
        """Called by run_workload at the start of every timed region (inside the captured graph, so that each replay starts
        with it): a fresh cohort.  A finished patient costs nothing (the kernel returns at once), so a region that ran on into
        days where everyone is dead would report a throughput that no live cohort has."""
        self.env.reset(self.init)
        self.day = 0

    def step(self):
        if self.day and self.day % self.COHORT_DAYS == 0:      # long regions: the next cohort
            self.env.reset(self.init)
        out = self.env.step(self.actions[self.i % self.actions.shape[0]])
        self.i += 1
        self.day += 1
        return out

    def units_per_step(self):
        return self.B

    def algorithmic_bytes_per_step(self):      # row read + written (f64) + per-patient scalars
        return (16 * self.env.nx + 96) * self.B

    compulsory_bytes_per_step = algorithmic_bytes_per_step

    def config(self):
        return {"workload": self.name, "env": "PDEControlGym-BrainTumor1D", "nodes": self.env.nx, "batch_per_gpu": self.B,
                "substeps_per_env_step": 1, "reward": "BrainTumorReward", "parallelism": "independent instances, no collective"}


from bench_ns2d import (NavierStokesC4, NavierStokesC4B4096, NavierStokesC4B4096F64, NavierStokesC4F64, NavierStokesC5,  # noqa: E402
                        NavierStokesC5F64, NavierStokesExample)
WORKLOADS["parabolic_c2_policy_loop"] = ParabolicPolicyLoop
WORKLOADS["parabolic_c2_rollout"] = ParabolicRollout
WORKLOADS["parabolic_c2_policy_loop_256"] = ParabolicPolicyLoop256
WORKLOADS["parabolic_c2_rollout_256"] = ParabolicRollout256
WORKLOADS["parabolic_c2_open_loop_rollout"] = ParabolicOpenLoopRollout
WORKLOADS["parabolic_c2_s1"] = ParabolicS1
WORKLOADS["parabolic_c2_s1_open_loop_rollout"] = ParabolicS1OpenLoopRollout
WORKLOADS["parabolic_c2_s1_rollout"] = ParabolicS1Rollout
WORKLOADS["ns2d_c4"] = NavierStokesC4
WORKLOADS["ns2d_c4_f64"] = NavierStokesC4F64
WORKLOADS["ns2d_c4_b4096"] = NavierStokesC4B4096
WORKLOADS["ns2d_c5"] = NavierStokesC5
WORKLOADS["ns2d_c5_f64"] = NavierStokesC5F64
WORKLOADS["ns2d_c4_f64_b4096"] = NavierStokesC4B4096F64
WORKLOADS["ns2d_example"] = NavierStokesExample
WORKLOADS["traffic_arz"] = TrafficARZ
WORKLOADS["traffic_arz_rollout"] = TrafficARZRollout
WORKLOADS["brain_tumor"] = BrainTumor


# Which kernel sources each workload's launches are compiled from: the committed PMC counters of a workload are stamped with the
# fingerprint of these files (build.sources_fingerprint) when they are collected, and roofline_block flags them as stale when the
# tree has moved on ("roofline counters are not tied to the binary").
_SRC_1D = ["pdegym_1d.hip", "pdegym_1d_body.h", "pdegym_common.h"]                                       # the step kernels
_SRC_1D_ROLL = ["pdegym_1d_rollout.hip", "pdegym_1d_body.h", "pdegym_common.h", "pdegym_policy.h", "pdegym_mlp_tile.h"]   # the rollout kernels
_SRC_NS = ["pdegym_ns2d.hip", "pdegym_ns_common.h", "pdegym_common.h"]
KERNEL_SOURCES = {
    "parabolic_c2": _SRC_1D, "transport_c3": _SRC_1D, "burgers_c3": _SRC_1D, "parabolic_c2_rollout": _SRC_1D_ROLL,
    "parabolic_c2_open_loop_rollout": _SRC_1D_ROLL, "parabolic_c2_s1": _SRC_1D, "parabolic_c2_s1_open_loop_rollout": _SRC_1D_ROLL,
    "parabolic_c2_s1_rollout": _SRC_1D_ROLL, "parabolic_c2_policy_loop": _SRC_1D + ["pdegym_mlp.hip", "pdegym_mlp_tile.h"],
    "parabolic_c2_policy_loop_256": _SRC_1D + ["pdegym_mlp.hip", "pdegym_mlp_tile.h"], "parabolic_c2_rollout_256": _SRC_1D_ROLL,
    "ns2d_c4": _SRC_NS, "ns2d_c4_b4096": _SRC_NS, "ns2d_c4_f64": _SRC_NS, "ns2d_c4_f64_b4096": _SRC_NS, "ns2d_example": _SRC_NS,
    "ns2d_c5": _SRC_NS + ["pdegym_ns256.hip", "pdegym_ns256_rows.h"],
    "ns2d_c5_f64": _SRC_NS + ["pdegym_ns256_f64.hip", "pdegym_ns256_rows.h"],
    "traffic_arz": ["pdegym_traffic.hip", "pdegym_common.h", "pdegym_policy.h", "pdegym_mlp_tile.h"],
    "traffic_arz_rollout": ["pdegym_traffic.hip", "pdegym_common.h", "pdegym_policy.h", "pdegym_mlp_tile.h"],
    "brain_tumor": ["pdegym_tumor.hip", "pdegym_common.h"],
}


def kernel_stamp(workload_key):
    from pdecontrolgym_amd import build
    files = [f for f in KERNEL_SOURCES.get(workload_key, []) if os.path.exists(os.path.join(build.CSRC, f))]
    return build.sources_fingerprint(files) if files else None


CLOCK_GHZ = 2.4          # MI355X peak shader clock (MI355X_MICROARCH.md); sustained clocks under VALU load are nearer 2.0
N_SIMD = 256 * 4         # 256 CUs x 4 SIMDs
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / 2.0     # one wave64 VALU instruction per SIMD every 2 cycles -> 1228.8 G wave-inst/s
REPEATS = 5              # median of 5 repeats of the K timed steps
WARM_SECONDS = 0.05      # untimed graph replays before the first timed region: at least this long ...
WARM_MAX_SECONDS = 1.0   # ... and until two consecutive replays agree to 1 %, but no longer than this


def profiled_counters(workload_key):
    """Per-env-step-batch counters of the last committed rocprofv3 collection (tools/profile_round.sh ->
    tools/summarize_profiles.py -> profiles/counters_latest.json): HBM bytes (FETCH_SIZE*2 + WRITE_SIZE, separate PMC passes)
    and SQ_INSTS_VALU summed over the kernels of one step.  bench.py itself never runs under the profiler; the counters
    belong to the default batch / sub-step count of the workload."""
    path = os.path.join(ROOT, "profiles", "counters_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d["workloads"].get(workload_key), d.get("source")
    except Exception:
        return None, None


def live_counters_from_csvs(paths, ksub):
    """rocprofv3's CSVs -> the per-step figures of the kernel whose name contains `ksub`: `paths["stats"]` = p_kernel_stats.csv of a
    `--kernel-trace --stats` pass, `paths[c]` = p_counter_collection.csv of a `--pmc c` pass for c in SQ_INSTS_VALU / FETCH_SIZE /
    WRITE_SIZE.  Per-launch averages; HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM / rocprofv3:
    gfx950's FETCH_SIZE tallies 128-byte reads at 64 bytes).  None if the kernel is missing or a pass saw fewer than 10 launches."""
    import csv
    out = {}
    for r in csv.DictReader(open(paths["stats"])):
        if ksub in r["Name"]:
            out.update(kernel=r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-80:],
                       calls=int(r["Calls"]), kernel_avg_ns=float(r["AverageNs"]), kernel_min_ns=float(r["MinNs"]))
            break
    else:
        return None
    for counter, scale, key in (("SQ_INSTS_VALU", 1.0, "valu_insts_per_step"), ("FETCH_SIZE", 2048.0, "_fetch"), ("WRITE_SIZE", 1024.0, "_write")):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(paths[counter]))
                if ksub in r["Kernel_Name"] and r["Counter_Name"] == counter]
        if len(vals) < 10:
            return None
        out[key] = sum(vals) / len(vals) * scale
    out["hbm_bytes_per_step"] = out.pop("_fetch") + out.pop("_write")
    return out


def live_counters(workload_key, budget_s=210.0, first_timeout_s=150.0, timeout_s=45.0):
    """Counters of THIS run's headline kernel, collected on THIS box: before the parent process touches the
    GPU, four fresh child processes run the headline workload under rocprofv3 -- `--kernel-trace --stats` (the kernel's average
    duration), then `--pmc SQ_INSTS_VALU`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` in passes of their own (never combined with a
    sys / hip trace; `python3` directly after `--`).  Returns {"kernel", "kernel_avg_ns", "calls", "valu_insts_per_step",
    "hbm_bytes_per_step", ...} or None (any failure, profiler missing, time budget spent: the caller falls back to the committed
    counters).  Never retried, and never run from a process that has used the GPU.  Timeouts: the first child may meet a cold box (the
    first `import torch` of a fresh image can take a minute or two), the others ~5 s each on a warm one; a run in which the profiler hangs
    outright still finishes within the "few minutes" of the bench contract."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    ksub = {"parabolic_c2": "step1d_kernel"}.get(workload_key)
    if ksub is None:
        return None
    t_begin = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="pdegym_live_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = ["python3", os.path.abspath(__file__), "--workload", workload_key, "--no-also", "--no-cpu-baseline", "--no-live-counters",
             "--steps", "30", "--warmup", "5", "--repeats", "1"]
    out = {"collected": "rocprofv3 children of this bench.py run, before the parent initialised the GPU"}

    def run(tag, opts):
        left = budget_s - (time.perf_counter() - t_begin)
        if left <= 5.0:
            raise TimeoutError("live-counter time budget spent")
        d = os.path.join(tmp, tag)
        cmd = [rocprof] + opts + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + child
        # own session: on a timeout the WHOLE group goes (rocprofv3 is a launcher; an orphaned grandchild would keep the GPU busy
        # under the timed regions that follow)
        pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp", start_new_session=True)
        try:
            rc = pr.wait(timeout=min(left, first_timeout_s if tag == "stats" else timeout_s))
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except OSError:
                pass
            pr.wait()
            raise
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd[0])
        for base, _, files in os.walk(d):          # (rocprofv3 may add a host-name level under -d)
            for f in files:
                if f in ("p_kernel_stats.csv", "p_counter_collection.csv"):
                    return os.path.join(base, f)
        raise FileNotFoundError(tag)
    try:
        paths = {"stats": run("stats", ["--kernel-trace", "--stats"])}
        for counter in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
            paths[counter] = run(counter, ["--pmc", counter, "--kernel-trace"])
        got = live_counters_from_csvs(paths, ksub)
        if got is None:
            return None
        out.update(got)
        out["seconds"] = time.perf_counter() - t_begin
        return out
    except Exception as ex:
        sys.stderr.write(f"live counters unavailable ({ex!r}); the roofline block uses the committed counters\n")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _timed(fn, world):
    """barrier + synchronize on both sides; HIP events on the launch stream bracket the same region."""
    import torch
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _barrier(world)
    t0 = time.perf_counter()
    ev0.record()
    fn()
    ev1.record()
    _barrier(world)
    return time.perf_counter() - t0, ev0.elapsed_time(ev1)


def run_workload(wl, steps, warmup, world, graph=False, repeats=REPEATS):
    """W untimed warm-up steps, then `repeats` timed regions of EXACTLY `steps` steps each (barrier + synchronize on both
    sides of every region).  Returns a dict: the median region (seconds, max over ranks), this rank's own median, every
    region's time, the per-step duration from HIP events over the median region, and the median duration of isolated
    single launches timed one by one afterwards.  graph=True: the `steps` launches are captured once into a hipGraph and
    each region is one replay (no Python between launches)."""
    import torch
    from pdecontrolgym_amd.sharding import max_over_ranks
    extra = min(steps, 50)
    begin = getattr(wl, "begin_region", lambda: None)     # workloads with finite episodes and no auto-reset restart here
    wl.prepare(warmup + ((repeats + 2) * steps) + extra)
    for _ in range(warmup):
        wl.step()
    regions = []
    if graph:
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            wl.step()                       # warm the side stream
            wl.i -= 1
            with torch.cuda.graph(g, stream=side):
                begin()
                for _ in range(steps):
                    wl.step()
        torch.cuda.current_stream().wait_stream(side)
        # untimed replays (graph upload, clocks, caches) until at least WARM_SECONDS have passed: short regions used to be
        # timed while the box was still speeding up (five consecutive regions fell monotonically by 6-10 %)
        # ... and, round 5, until two consecutive replays agree to 1 % (at most WARM_MAX_SECONDS): after a different workload the
        # clocks of a box settle over hundreds of milliseconds, and a region of 10 ms replays (the NS workloads) was still drifting by
        # 8-22 % across its five timed regions when only the 50 ms rule applied (profiles/r05d_bench_also.json: timed_regions_s)
        t_warm = time.perf_counter()
        last, agree = None, 0
        while True:
            t_r = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            now = time.perf_counter()
            dur = now - t_r
            agree = agree + 1 if (last is not None and abs(dur - last) <= 0.01 * dur) else 0
            last = dur
            if now - t_warm >= WARM_SECONDS and (agree >= 2 or now - t_warm >= WARM_MAX_SECONDS):
                break
        for _ in range(repeats):
            regions.append(_timed(g.replay, world))
    else:
        def loop():
            begin()
            for _ in range(steps):
                wl.step()
        for _ in range(repeats):
            regions.append(_timed(loop, world))
    # the only communication of a multi-GPU run (no data-path collective): MAX over ranks of every region's time
    maxes = [max_over_ranks(r[0], device="cuda") for r in regions]
    med = sorted(range(repeats), key=lambda k: maxes[k])[repeats // 2]
    el, (el_local, ev_ms) = maxes[med], regions[med]
    # per-launch duration with HIP events on the launch stream (outside the timed regions)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(extra)]
    begin()
    for a, b in evs:
        a.record()
        wl.step()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return {"seconds": el, "seconds_this_rank": el_local, "step_ms_events": ev_ms / steps, "isolated_step_ms_median": ms[len(ms) // 2],
            "all_regions_s": maxes}


def vecenv_host_rate(device, B=4096, steps=60, warmup=8):
    """The SB3-facing path at the C2 shape: ``PDEVecEnv.step(numpy actions)`` -> NumPy observations / rewards / dones / infos,
    host round trip included (what ``PPO("MlpPolicy", venv).learn()`` drives; reference caller
    examples/transportPDE/transport1Dppo.py:77-90).  PCIe-inclusive, so it is reported beside the headline, never as it."""
    import numpy as np
    import pde_control_gym
    from pde_control_gym.src import TunedReward1D
    nx, S = Parabolic1D.nx, Parabolic1D.S
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    beta = (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)
    rng = np.random.default_rng(0)
    p = {"T": 1000 * S * dt, "dt": dt, "X": 1, "dx": dx, "reward_class": TunedReward1D(1000 * S, -1e3, 3e2), "normalize": True,
         "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None, "sensing_noise_func": None,
         "limit_pde_state_size": True, "max_state_value": 1e10, "max_control_value": 20, "control_sample_rate": S * dt,
         "batched_reset_func": lambda idx, nx_: (rng.uniform(1, 10, (len(idx), 1)).astype(np.float32) * np.ones((1, nx_ + 1), np.float32),
                                                 np.tile(beta, (len(idx), 1)))}
    acts = np.random.default_rng(1).uniform(-1, 1, (warmup + steps, B, 1)).astype(np.float32)
    venv = pde_control_gym.make_vec("PDEControlGym-ReactionDiffusionPDE1D", num_envs=B, device=str(device), **p)
    venv.reset()
    venv.enable_fused_auto_reset()
    for k in range(warmup):
        venv.step(acts[k])
    t0 = time.perf_counter()
    for k in range(steps):
        obs, rew, dones, infos = venv.step(acts[warmup + k])       # (results dropped every step, as SB3's loop does after copying)
    el = (time.perf_counter() - t0) / steps
    kept = []
    t0 = time.perf_counter()
    for k in range(steps):
        kept.append(venv.step(acts[warmup + k])[0])                # a caller that KEEPS every observation (reference-style list)
    el_keep = (time.perf_counter() - t0) / steps
    del kept
    # the path every caller gets on an interpreter without exact reference counts, under a tool that holds hidden references, or with
    # copy_outputs=True: no recycling, every result a plain NumPy copy staged through one pinned buffer (VERDICT r4 "weak" 6)
    venv.copy_outputs = True
    for k in range(warmup):
        venv.step(acts[k])
    t0 = time.perf_counter()
    for k in range(steps):
        obs, rew, dones, infos = venv.step(acts[warmup + k])
    el_copy = (time.perf_counter() - t0) / steps
    venv.copy_outputs = None
    return {"value": B / el, "unit": "env-steps/s", "us_per_step": el * 1e6, "batch": B,
            "host_bytes_per_step": int(obs.nbytes + rew.nbytes + dones.nbytes + acts[0].nbytes),
            "caller_keeps_every_observation": {"value": B / el_keep, "us_per_step": el_keep * 1e6},
            "copy_outputs_true": {"value": B / el_copy, "us_per_step": el_copy * 1e6},
            "note": "PDEVecEnv.step: numpy actions in, numpy observations / rewards / dones / infos out; the arrays are views of pinned "
                    "staging buffers recycled by reference count (never overwritten while the caller holds them), one stream "
                    "synchronisation per step; PCIe-inclusive (never the headline value)"}


def hbm_probe(device):
    """The measured HBM yardstick of THIS box, printed beside the 8 TB/s specification: hand-written float4 copy / read / fill
    (tools/hbm_probe.hip, 1 GiB, best of two grid sizes x plain / non-temporal).  ~0.1 s."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hbm_probe as hp
    r = hp.measure(device, quick=True)
    return {k: r[k] for k in ("copy_GBps", "read_GBps", "fill_GBps", "at_MiB", "kernel")}


def roofline_block(wl, key, step_ms, default_config, live=None):
    """The resource that binds the step and how close the step is to it.  Two candidates, both reported:
      * VALU issue: SQ_INSTS_VALU of one step (profiled) x 2 cycles / (1024 SIMDs x step time x 2.4 GHz);
      * HBM: PMC bytes of one step (profiled) / step time against the 8 TB/s specification.
    `bound` names the larger fraction.  The SURVEY section 8d streaming-model figure (algorithmic bytes / time) is kept as
    `effective_*`: the fused kernels keep state on chip across sub-steps / sweeps, so it may exceed the HBM peak and is NOT
    a utilisation."""
    t = step_ms * 1e-3
    alg = wl.algorithmic_bytes_per_step()
    ctr, src = profiled_counters(key) if default_config else (None, None)
    if live and default_config:          # counters collected by this very run (live_counters) take the place of the committed ones
        ctr = dict(live, kernel_stamp=kernel_stamp(key), kernels={live.get("kernel"): {"avg_ns": live.get("kernel_avg_ns")}})
        src = live["collected"]
    valu = hbm = None
    if ctr and ctr.get("valu_insts_per_step"):
        a = ctr["valu_insts_per_step"] / t / 1e9
        valu = {"wave_insts_per_step": ctr["valu_insts_per_step"], "achieved": a, "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s",
                "frac": a / VALU_PEAK_GINST}
        if ctr.get("f64_insts_per_step") is not None:
            # a float64 add / mul / fma occupies the SIMD for 4 cycles (16 lanes per clock) where everything else takes 2: the share
            # of the SIMDs' time the instruction stream needs = 2 x (all + float64 ones) cycles / (1024 SIMDs x step time x 2.4 GHz).
            # `frac` above counts every instruction once (comparable across workloads); this one is the truer figure for float64 kernels
            valu["f64_wave_insts_per_step"] = ctr["f64_insts_per_step"]
            valu["frac_f64_weighted"] = (ctr["valu_insts_per_step"] + ctr["f64_insts_per_step"]) / t / 1e9 / VALU_PEAK_GINST
    if ctr and ctr.get("hbm_bytes_per_step"):
        a = ctr["hbm_bytes_per_step"] / t / 1e9
        hbm = {"bytes_per_step": ctr["hbm_bytes_per_step"], "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS}
    cands = [(v["frac"], n, v) for n, v in (("valu_issue", valu), ("hbm", hbm)) if v]
    out = {}
    if cands:
        _, name, best = max(cands)
        out = {"bound": name, "achieved": best["achieved"], "peak": best["peak"], "unit": best["unit"], "frac": best["frac"]}
    else:       # no profile for this configuration: only the effective figure is available
        eff = alg / t / 1e9
        out = {"bound": "hbm", "achieved": eff, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
               "note": "no committed counters for this batch / sub-step count: achieved is the EFFECTIVE streaming-model bandwidth"}
    out["traffic"] = hbm["bytes_per_step"] if hbm else None
    if ctr and ctr.get("round"):
        src = f"round {ctr['round']}: " + (src or "")
    if ctr:
        # the counters were collected from kernels built from sources with this fingerprint; if the tree (or, on a GPU box, the
        # shipped library's own stamp) has moved on since, the fractions below describe an OLDER kernel's instruction / byte counts
        from pdecontrolgym_amd import build
        now = kernel_stamp(key)
        out["counters_kernel_stamp"] = ctr.get("kernel_stamp")
        lib = build.library_stamp()
        out["counters_stale"] = bool(ctr.get("kernel_stamp") != now or (lib != "" and lib != build._fingerprint()))
    out["counters_live"] = bool(live and default_config)
    if live and default_config:
        out["kernel_avg_ns"] = live.get("kernel_avg_ns")        # rocprofv3 --stats average of the dominant kernel, this run, this box
    out.update({"valu_issue": valu, "hbm": hbm, "step_ms": step_ms, "counters_source": src,
                "kernels_per_step": (ctr or {}).get("kernels"),
                "effective_streaming_GBps": alg / t / 1e9, "effective_streaming_frac_of_hbm_peak": alg / t / 1e9 / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_step": alg, "compulsory_bytes_per_step": wl.compulsory_bytes_per_step()})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="parabolic_c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="instances per GPU (default: the BASELINE batch)")
    ap.add_argument("--substeps", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads in the default run")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not collect the headline kernel's counters with rocprofv3 children first (committed counters are used)")
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed regions of K steps each; the median is reported")
    ap.add_argument("--eager", action="store_true",
                    help="one Python call per launch instead of replaying the K timed launches from one captured hipGraph")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--seed", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:                      # child of cpu_baseline_report: NumPy only, prints env-steps/s
        print(cpu_port_rate(args.cpu_worker, args.cpu_seconds, args.seed)[0])
        return
    # Live counters + CPU baseline first: their children must be started before this process initialises the GPU
    live = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_live_counters and not args.no_also and not args.batch
            and not args.substeps and args.workload == "parabolic_c2"):
        live = live_counters(args.workload)
    cpu_rep = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        cpu_rep = cpu_baseline_report(args.workload, args.cpu_seconds)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to benchmark)")
    rank, local, world = _dist_setup(args.gpus)
    device = torch.device("cuda", local)
    kw = {}
    if args.batch:
        kw["B"] = args.batch
    if args.substeps:
        kw["S"] = args.substeps
    wl = WORKLOADS[args.workload](device, 1234 + rank, **kw)
    use_graph = not args.eager
    try:
        res = run_workload(wl, args.steps, args.warmup, world, graph=use_graph, repeats=args.repeats)
    except Exception as ex:                 # capture unsupported in this environment -> time the eager loop instead
        if not use_graph:
            raise
        sys.stderr.write(f"hipGraph capture failed ({ex!r}); falling back to the eager loop\n")
        use_graph = False
        torch.cuda.synchronize()
        wl = WORKLOADS[args.workload](device, 1234 + rank, **kw)
        res = run_workload(wl, args.steps, args.warmup, world, graph=False, repeats=args.repeats)
    el = res["seconds"]
    value = wl.units_per_step() * args.steps * world / el
    default_config = not args.batch and not args.substeps
    out = {
        "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
        "config": dict(wl.config(), launch="hipGraph replay of the K timed launches" if use_graph else "eager (one Python call per launch)",
                       timing=f"median of {args.repeats} regions of exactly K steps, each bracketed by barrier + synchronize; max over ranks"),
        "timed_regions_s": res["all_regions_s"],
        "roofline": roofline_block(wl, args.workload, res["step_ms_events"], default_config, live),
    }
    out["roofline"]["isolated_step_ms_median"] = res["isolated_step_ms_median"]
    if world > 1:
        # per-rank figures so that a scaling record can be cross-checked: every rank steps its own full per-GPU batch
        # (weak scaling), so each per-rank value should equal the N=1 line of the same box
        import torch.distributed as dist
        mine = torch.tensor([wl.units_per_step() * args.steps / res["seconds_this_rank"]], dtype=torch.float64,
                            device="cpu" if dist.get_backend() == "gloo" else device)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [float(v.item()) for v in allv]
        out["per_rank_env_steps_per_s"] = per_rank
        out["n1_equivalent"] = {"value": sum(per_rank) / world, "note": "mean per-rank rate = what one GPU of this node does alone; "
                                "value / (n_gpus * this) is the scaling efficiency the driver computes from its own N=1 run"}
        out["process_group"] = dist.get_backend()
    if cpu_rep is not None:
        out["cpu_baseline"] = cpu_rep
    also = None
    if world > 1 and not args.no_also and args.workload == "parabolic_c2" and not args.batch:
        # BASELINE configs[4] is the one configuration DEFINED on 8 GPUs (NavierStokes2D 256 x 256, 4096 instances sharded 512 per
        # GPU): every rank steps its own 512 instances (no data-path collective), regions bracketed by the barrier, time = MAX over
        # ranks.  Node totals + the per-rank rates; float32 and the reference's own float64.
        import torch.distributed as dist
        multi = {}
        for name in MULTI_GPU_SECONDARY:
            try:
                w2 = WORKLOADS[name](device, 99 + rank)
                n2 = 20
                r2 = run_workload(w2, n2, 5, world, graph=use_graph, repeats=3)
                mine = torch.tensor([w2.units_per_step() * n2 / r2["seconds_this_rank"]], dtype=torch.float64,
                                    device="cpu" if dist.get_backend() == "gloo" else device)
                allv = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allv, mine)
                multi[name] = {"value": w2.units_per_step() * n2 * world / r2["seconds"], "unit": "env-steps/s",
                               "ms_per_step": r2["seconds"] / n2 * 1e3, "dtype": w2.dtype, "timed_regions_s": r2["all_regions_s"],
                               "per_rank_env_steps_per_s": [float(v.item()) for v in allv], "instances_per_gpu": w2.B,
                               "instances_whole_node": w2.B * world, "config": w2.config(),
                               "roofline": {"bound": None, "frac": None, "step_ms": r2["step_ms_events"]}}
                del w2
            except Exception as ex:          # every rank takes the same path: a failure here fails on all of them alike
                multi[name] = {"error": repr(ex)}
        if rank == 0:
            also = multi
            out["also"] = also
    if rank == 0 and world == 1 and not args.no_also and args.workload == "parabolic_c2" and not args.batch:
        also = {}
        try:            # first: a process that has built and dropped nineteen workloads hands out host memory far more slowly
            also["vecenv_host"] = vecenv_host_rate(device)
        except Exception as ex:
            also["vecenv_host"] = {"error": repr(ex)}
        try:            # the single-environment drop-in face (batch of one) beside the un-batched NumPy oracle (bench_single.py)
            from bench_single import single_env_block
            also["single_env"] = single_env_block(device)
        except Exception as ex:
            also["single_env"] = {"error": repr(ex)}
        try:
            also["hbm_probe"] = hbm_probe(device)
        except Exception as ex:
            also["hbm_probe"] = {"error": repr(ex)}
        for name, cls in WORKLOADS.items():
            if name == args.workload:
                continue
            try:
                w2 = cls(device, 99)
                n2 = max(40, args.steps // 2) if name not in ("ns2d_c4_b4096", "ns2d_c5", "ns2d_example", "ns2d_c5_f64", "ns2d_c4_f64_b4096") else 20
                r2 = run_workload(w2, n2, max(5, args.warmup // 2), 1, graph=use_graph, repeats=REPEATS)
                rf = roofline_block(w2, name, r2["step_ms_events"], True)
                also[name] = {"value": w2.units_per_step() * n2 / r2["seconds"], "unit": "env-steps/s", "ms_per_step": r2["seconds"] / n2 * 1e3,
                              "dtype": w2.dtype, "timed_regions_s": r2["all_regions_s"],
                              "roofline": dict({k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "step_ms", "counters_stale")},
                                               valu_issue_frac=(rf.get("valu_issue") or {}).get("frac"), hbm_frac=(rf.get("hbm") or {}).get("frac"),
                                               valu_issue_frac_f64_weighted=(rf.get("valu_issue") or {}).get("frac_f64_weighted")),
                              "config": w2.config()}
                del w2
            except Exception as ex:  # keep the headline line alive
                also[name] = {"error": repr(ex)}
        out["also"] = also
    if rank == 0:
        # The LAST stdout line is what the driver parses: it stays under 4 KB (the 23 KB line of round 4 was not
        # parsed).  Everything else -- per-region times, the full roofline dicts, every secondary workload -- goes to
        # bench_also.json next to this script (and to gpurun_out/ when that exists).
        # (only the full default run owns bench_also.json; a --workload X / --no-also run writes bench_<workload>.json instead, so
        # that tools looping over workloads do not replace "the full result of the run" the printed line points to)
        fname = "bench_also.json" if also is not None else f"bench_{args.workload}.json"
        for path in (os.path.join(ROOT, fname), os.path.join(ROOT, "gpurun_out", fname)):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, "w") as fh:
                        json.dump(out, fh, indent=1)
            except OSError:
                pass
        print(final_line(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


SECONDARY = ("ns2d_c4_b4096", "ns2d_c4_f64_b4096", "transport_c3", "ns2d_c4", "ns2d_c4_f64", "ns2d_c5", "ns2d_c5_f64",
             "parabolic_c2_s1_open_loop_rollout", "traffic_arz")      # the other BASELINE configs (+ the two kernels VERDICT r4 names)
MULTI_GPU_SECONDARY = ("ns2d_c5", "ns2d_c5_f64")    # measured rank-locally when world > 1 (BASELINE configs[4]: 512 instances per GPU)
MAX_LINE = 4096


def _r(x, sig=6):
    """Floats to `sig` significant digits (the full-precision numbers are in bench_also.json)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{sig}g}")


def _committed_kernel_avg(rf):
    """rocprofv3's average duration of the step's dominant (longest) kernel in the committed counters, if they carry one."""
    ks = [k.get("avg_ns") for k in (rf.get("kernels_per_step") or {}).values() if isinstance(k, dict) and k.get("avg_ns")]
    return max(ks) if ks else None


def final_line(out):
    """The one line the driver parses, from the full result dict: headline + roofline + cpu_baseline + a compact `secondary`
    list; < MAX_LINE bytes whatever the workload names and notes grow to (tests/test_bench_line.py)."""
    rf = out.get("roofline") or {}
    cfg = out.get("config") or {}
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {k: cfg[k] for k in ("workload", "env", "nx", "ny", "nodes", "batch_per_gpu", "substeps_per_env_step",
                                          "jacobi_sweeps_per_step", "env_steps_per_launch", "reward", "launch", "parallelism") if k in cfg}
    line["roofline"] = {
        "bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _r(rf.get("frac"), 4),
        "traffic": _r(rf.get("traffic")), "algorithmic_bytes_per_step": rf.get("algorithmic_bytes_per_step"),
        # step_event_ns: this run's HIP-event time per step over the timed region (what `achieved` / `frac` divide by);
        # kernel_avg_ns: rocprofv3 --stats average duration of the dominant kernel -- live (counters_live) or of the committed round
        "step_event_ns": _r(rf["step_ms"] * 1e6) if rf.get("step_ms") else None,
        "kernel_avg_ns": _r(rf.get("kernel_avg_ns") or _committed_kernel_avg(rf)),
        "counters_live": rf.get("counters_live"),
        "valu_wave_insts_per_step": _r((rf.get("valu_issue") or {}).get("wave_insts_per_step")),
        "counters_round": "live" if rf.get("counters_live") else ((rf.get("counters_source") or "").split(":")[0].replace("round ", "") or None),
        "counters_stale": rf.get("counters_stale"),
        "hbm_frac": _r((rf.get("hbm") or {}).get("frac"), 4), "valu_issue_frac": _r((rf.get("valu_issue") or {}).get("frac"), 4),
        "hbm_copy_measured_GBps": _r(((out.get("also") or {}).get("hbm_probe") or {}).get("copy_GBps"), 4),
    }
    cb = out.get("cpu_baseline")
    if cb:
        c = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
             "sample": (cb.get("sample") or "")[:160]}
        ac = cb.get("all_cores") or {}
        if "value" in ac:
            c["all_cores"] = {"value": _r(ac["value"]), "logical_cpus": ac.get("logical_cpus")}
        line["cpu_baseline"] = c
    if "per_rank_env_steps_per_s" in out:
        line["per_rank_env_steps_per_s"] = [_r(v) for v in out["per_rank_env_steps_per_s"]]
        line["n1_equivalent"] = _r(out["n1_equivalent"]["value"])
        line["process_group"] = out.get("process_group")
    also = out.get("also")
    if also:
        sec = []
        for name in SECONDARY:
            a = also.get(name)
            if not a or "error" in a:
                if a:
                    sec.append({"name": name, "error": a["error"][:60]})
                continue
            r2 = a.get("roofline") or {}
            e = {"name": name, "value": _r(a["value"]), "ms_per_step": _r(a["ms_per_step"], 5), "dtype": a.get("dtype"),
                 "bound": r2.get("bound"), "frac": _r(r2.get("frac"), 3)}
            if "per_rank_env_steps_per_s" in a:          # an N > 1 run: node total in `value`, every rank's own rate beside it
                e["per_rank"] = [_r(v, 4) for v in a["per_rank_env_steps_per_s"]]
                e["instances_per_gpu"] = a.get("instances_per_gpu")
            sec.append(e)
        line["secondary"] = sec
        vh = also.get("vecenv_host") or {}
        if "value" in vh:
            line["vecenv_host_env_steps_per_s"] = _r(vh["value"])
        se = also.get("single_env") or {}
        if se and "error" not in se:
            # microseconds per env.step() of ONE environment through the drop-in face: [GPU (history kept, the default), GPU without
            # history, un-batched NumPy oracle]; the reference's callers (DummyVecEnv(n=1)) drive exactly this path
            line["single_env_us_per_step"] = {k: [_r(v["gpu"]["us_per_step"], 4), _r((v.get("gpu_no_history") or v["gpu"])["us_per_step"], 4),
                                                  _r(v["numpy"]["us_per_step"], 4)]
                                              for k, v in se.items() if isinstance(v, dict) and "gpu" in v and "numpy" in v}
            line["single_env_gpu_wins_above_substeps"] = _r((se.get("crossover") or {}).get("substeps_above_which_gpu_wins"), 3)
        line["details"] = "bench_also.json"
    s = json.dumps(line, separators=(",", ":"))
    while len(s) >= MAX_LINE and line.get("secondary"):      # never expected; keeps the contract if names / notes grow
        line["secondary"].pop()
        s = json.dumps(line, separators=(",", ":"))
    if len(s) >= MAX_LINE:
        line["cpu_baseline"] = {k: v for k, v in (line.get("cpu_baseline") or {}).items() if k != "sample"}
        line["config"] = {"workload": str(cfg.get("workload"))[:200]}
        s = json.dumps(line, separators=(",", ":"))
    return s


if __name__ == "__main__":
    main()
