"""Whole-job throughput of N independent replicas (one process per GPU): the utterances all ranks processed
divided by the slowest rank's time.  The hot path is forward-only, utterances are independent, so ranks
exchange nothing but this one timing reduction (RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations


def job_throughput(elapsed_s: float, units: float, device=None, force: bool = False):
    """-> (units_per_second over all ranks, max elapsed).  Works with or without an initialised process group; force: issue the
    two reductions also in a group of ONE rank (rehearsal of the N-rank path on one GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return units / elapsed_s, elapsed_s
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u[0]) / float(t[0]), float(t[0])


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of a global batch for this rank (equal shards up to a remainder of one)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
