"""N>1 path on CPU: world_size-2 gloo run.  Each rank steps its contiguous shard of a globally generated batch
(host logic through the oracle test double); the concatenation of the shards must equal the single-process result
bitwise, and the timing reduction must return the max over ranks.  No data-path collective exists to test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pdecontrolgym_amd.sharding import gather_instances, max_over_ranks, shard_bounds


def test_shard_bounds_cover_the_batch():
    for total in (1, 7, 512, 4096, 4097):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def _global_batch(B, n):
    rng = np.random.default_rng(42)
    x = np.linspace(0, 1, n)
    init = (rng.uniform(1, 10, (B, 1)) * np.ones((1, n))).astype(np.float32)
    beta = (50 * np.cos(rng.uniform(7.5, 8.5, (B, 1)) * np.arccos(x))).astype(np.float32)
    acts = rng.uniform(-1, 1, (4, B)).astype(np.float32)
    return init, beta, acts


def _run_shard(lo, hi, init, beta, acts):
    from pdecontrolgym_amd import _native as N
    from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec
    from tests.fake_backend import FakeBackend
    nx = 64
    dx = 1.0 / nx
    dt = 0.25 * dx * dx
    env = PDEBatch1D("parabolic", 400 * dt, dt, 1, dx, 20 * dt, limit_pde_state_size=True,
                     reward=RewardSpec(N.REWARD_TUNED1D, 400, -1e3, 3e2), num_envs=hi - lo, device="cpu", backend=FakeBackend())
    env.reset(torch.tensor(init[lo:hi]), torch.tensor(beta[lo:hi]))
    for a in acts:
        obs, r, te, tr = env.step(torch.tensor(a[lo:hi]))
    return env.u.clone(), r.clone()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n = 10, 65
    init, beta, acts = _global_batch(B, n)
    lo, hi = shard_bounds(B, rank, world)
    u, r = _run_shard(lo, hi, init, beta, acts)
    full_u = gather_instances(u, B)
    full_r = gather_instances(r, B)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((full_u.numpy(), full_r.numpy(), t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_single_process_bitwise():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full_u, full_r, t = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, n = 10, 65
    init, beta, acts = _global_batch(B, n)
    u1, r1 = _run_shard(0, B, init, beta, acts)
    np.testing.assert_array_equal(full_u, u1.numpy())
    np.testing.assert_array_equal(full_r, r1.numpy())
    assert t == 2.0
