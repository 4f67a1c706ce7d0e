"""Multi-GPU layout: environment instances are independent, so a node of N GPUs runs N replicas of the
stepper, each owning a contiguous block of the instance axis.  There is NO collective on the step path
(SURVEY.md section 8e); torch.distributed (RCCL on GPUs, gloo in CPU tests) is only used for the timing
barrier / max-over-ranks and for optional gathering of per-instance scalars for logging.
"""
from __future__ import annotations


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous block split of ``total`` instances: rank r owns [lo, hi). Remainders go to the first ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (elapsed seconds); identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_instances(local, total: int):
    """Concatenate per-rank blocks (made with shard_bounds) of a per-instance tensor on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
