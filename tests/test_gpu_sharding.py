"""N>1 path on the HIP backend with ONE GPU: two ranks share cuda:0 (the box has a single device; the driver's 8-GPU run
is not ours to launch).  Each rank steps its contiguous shard of a globally generated batch through libpdegym_hip.so; the
concatenation of the shards equals the single-process result bitwise for the 1D and the NS2D steppers, the process group
has two ranks (RCCL when it accepts two ranks on one device, gloo otherwise -- the step path has no collective either way),
and bench.py --gpus 2 produces a line with per-rank values."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PDEGYM_ROOT"])
from pdecontrolgym_amd.sharding import gather_instances, max_over_ranks, shard_bounds
from pdecontrolgym_amd import _native as N
from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
from pdecontrolgym_amd.batch2d import NSBatch2D
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
backend = "nccl"
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    assert int(t.item()) == world
except Exception as ex:
    try: dist.destroy_process_group()
    except Exception: pass
    backend = "gloo"
    dist.init_process_group("gloo")
def batch1d(B, n):
    rng = np.random.default_rng(42)
    x = np.linspace(0, 1, n)
    init = (rng.uniform(1, 10, (B, 1)) * np.ones((1, n))).astype(np.float32)
    beta = (50 * np.cos(rng.uniform(7.5, 8.5, (B, 1)) * np.arccos(x))).astype(np.float32)
    return init, beta, rng.uniform(-1, 1, (4, B)).astype(np.float32)
def run1d(lo, hi, init, beta, acts):
    dx = 1.0 / 256; dt = 0.25 * dx * dx
    env = PDEBatch1D("parabolic", 400 * dt, dt, 1, dx, 100 * dt, limit_pde_state_size=True, reward=RewardSpec(N.REWARD_TUNED1D, 400, -1e3, 3e2),
                     num_envs=hi - lo, device="cuda")
    env.reset(torch.tensor(init[lo:hi]), torch.tensor(beta[lo:hi]))
    for a in acts: obs, r, te, tr = env.step(torch.tensor(a[lo:hi]))
    return env.u.clone(), r.clone()
def batchns(B, n):
    rng = np.random.default_rng(7)
    return rng.uniform(-1, 1, (3, B, n, n)).astype(np.float32), rng.uniform(2, 4, (3, B)).astype(np.float32)
def runns(lo, hi, ic, acts, n):
    dx = 1.0 / (n - 1); dt = 0.2 * 0.5 * dx * dx / 0.1
    bc = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}
    env = NSBatch2D(8 * dt, dt, 1, dx, 1, dx, bc, np.zeros((8, n, n, 2), dtype=np.float32), 2.0 * np.ones(8, dtype=np.float32), gamma=0.1,
                    maximum_pressure_iteration=50, num_envs=hi - lo, device="cuda", dtype=torch.float32)
    env.reset(ic[0][lo:hi], ic[1][lo:hi], ic[2][lo:hi])
    for a in acts: obs, r, te = env.step(a[lo:hi])
    return obs.clone(), r.clone()
out = {"backend": backend, "world": dist.get_world_size()}
B, n = 1030, 257
init, beta, acts = batch1d(B, n)
lo, hi = shard_bounds(B, rank, world)
u, r = run1d(lo, hi, init, beta, acts)
dev = None if backend == "nccl" else "cpu"
gu = gather_instances(u if backend == "nccl" else u.cpu(), B); gr = gather_instances(r if backend == "nccl" else r.cpu(), B)
Bn, nn = 6, 128
ic, an = batchns(Bn, nn)
lo, hi = shard_bounds(Bn, rank, world)
o, rn = runns(lo, hi, ic, an, nn)
go = gather_instances(o if backend == "nccl" else o.cpu(), Bn); grn = gather_instances(rn if backend == "nccl" else rn.cpu(), Bn)
tmax = max_over_ranks(1.0 + rank, device="cuda")
if rank == 0:
    u1, r1 = run1d(0, B, init, beta, acts)
    o1, rn1 = runns(0, Bn, ic, an, nn)
    out.update(eq_u=bool(torch.equal(gu.cpu(), u1.cpu())), eq_r=bool(torch.equal(gr.cpu(), r1.cpu())), eq_obs=bool(torch.equal(go.cpu(), o1.cpu())),
               eq_rn=bool(torch.equal(grn.cpu(), rn1.cpu())), tmax=tmax)
    print("RESULT " + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(args, extra_env=None, timeout=600):
    env = dict(os.environ, PDEGYM_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(900)
def test_two_rank_shards_on_the_hip_backend_equal_single_process_bitwise(tmp_path):
    script = tmp_path / "shard_worker.py"
    script.write_text(WORKER)
    r = _launch([str(script)])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[-1][7:])
    assert out["world"] == 2 and out["backend"] in ("nccl", "gloo")
    assert out["eq_u"] and out["eq_r"] and out["eq_obs"] and out["eq_rn"], out
    assert out["tmax"] == 2.0


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_per_rank_values():
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--repeats", "3"],
                extra_env={"PDEGYM_BENCH_SHARE_GPU": "1"})
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert len(d["per_rank_env_steps_per_s"]) == 2 and d["process_group"] in ("nccl", "gloo")
    # value = all instances of both ranks / max-over-ranks time  <=  sum of the per-rank rates
    assert d["value"] <= sum(d["per_rank_env_steps_per_s"]) * 1.001
    assert d["value"] > 0.5 * min(d["per_rank_env_steps_per_s"])
    # ... and the configuration that is DEFINED on several GPUs (BASELINE configs[4]: NavierStokes2D 256 x 256, 512 instances per
    # GPU), float32 and float64: node total + every rank's own rate
    sec = {e["name"]: e for e in d["secondary"]}
    for name, dt in (("ns2d_c5", "f32"), ("ns2d_c5_f64", "f64")):
        e = sec[name]
        assert "error" not in e, e
        assert e["dtype"] == dt and e["instances_per_gpu"] == 512 and len(e["per_rank"]) == 2
        assert 0.5 * min(e["per_rank"]) < e["value"] <= sum(e["per_rank"]) * 1.001


@pytest.mark.timeout(900)
def test_bench_falls_back_to_gloo_when_rccl_cannot_initialise():
    """bench.py's RCCL -> gloo fallback for the timing barrier (the step path has no collective): with the nccl initialisation
    forced to fail the line is still printed, by a gloo process group."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--repeats", "3", "--no-also"],
                extra_env={"PDEGYM_BENCH_SHARE_GPU": "1", "PDEGYM_BENCH_FAIL_NCCL": "1"})
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["process_group"] == "gloo" and d["n_gpus"] == 2 and len(d["per_rank_env_steps_per_s"]) == 2
    assert "using gloo for the timing barrier" in r.stderr
