"""Data-parallel training of the hot path (BASELINE configs 4 and 5): one process per GPU, each rank a
contiguous shard of the global batch, and per step

  * Sync-CBN: train-mode ComplexBatchNormal takes its statistics over the WHOLE batch (reference
    model/complex_progress.py:132-143), so the per-channel moment sums the conv epilogue emits ([C][5] doubles) and the
    backward's reduction sums ([C][8] doubles) are all-reduced between the reduce kernel and the finalise kernel
    (``ops.BN_SYNC`` hook).  With equal shards the result equals the single-process step on the global batch.
  * one bucketed gradient all-reduce (sum / world) over RCCL after backward: 9.5-25 M fp32 values = 38-100 MB, i.e.
    0.2-1.1 ms on xGMI against a >= 100 ms step, so the collective is issued in few large buckets (default 128 MB:
    one for every shipped model) rather than overlapped piecemeal; buckets follow reverse registration order, which is
    the order backward produces the gradients in.

The reference itself is single-GPU (no torch.distributed anywhere); this is the MI355X-native addition north_star names.
``torch.distributed`` backend "nccl" is RCCL on ROCm; the CPU tests and single-GPU rehearsals use gloo, for which
device tensors are staged through host memory.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops


# Test hook (tests/test_gpu_rccl.py): with a process group of ONE rank the collectives are still issued instead of being
# skipped, so that the RCCL branch of every function below executes on a one-GPU box (a sum / average over one rank is the
# identity, the step's results must not change).
FORCE_COLLECTIVES = os.environ.get("IDV_DP_FORCE", "0") == "1"


def _world(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _active(group=None) -> bool:
    """Are collectives issued?  More than one rank -- or an initialised group of one with FORCE_COLLECTIVES."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce; gloo cannot take device tensors on every build, so they go through the host there."""
    if not _active(group):
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sync_moments(sums: torch.Tensor, group=None) -> int:
    """The ops.BN_SYNC hook: all-reduce per-channel moment sums, return the factor for the element count."""
    all_reduce_sum_(sums, group)
    return _world(group)


def enable_sync_bn(group=None):
    ops.BN_SYNC = (lambda sums: sync_moments(sums, group)) if _active(group) else None


def disable_sync_bn():
    ops.BN_SYNC = None


class GradAllReduce:
    """Bucketed gradient averaging for the parameters that require grad.

    Device parameters: per bucket ONE multi-tensor gather kernel (``idv_bucket_gather``: every ``.grad`` -> the flat bucket,
    absent gradients as zeros), the RCCL all-reduce (``ReduceOp.AVG``), and ONE scatter kernel back into the ``.grad``
    tensors -- or none at all: ``reduce(into_grads=False)`` leaves the averaged gradients in the bucket and
    ``optim.Adam.step(grad_bucket=red.bucket())`` reads them there.  Buckets are filled from the LAST registered parameters
    to the first (the order backward produces gradients in); inside a bucket the parameters keep registration order, so
    the one-bucket case (every shipped model: <= 100 MB against the 128 MB default) has the optimiser's own layout.
    CPU parameters (the gloo protocol test) take a plain torch loop."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 128 << 20, group=None):
        self.group = group
        ps = [p for p in params if p.requires_grad]
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(ps):
            n = p.numel() * p.element_size()
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur[::-1])
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self.buckets.append(cur[::-1])
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._tab = [None] * len(self.buckets)
        self._present = [None] * len(self.buckets)
        self._checked = [False] * len(self.buckets)
        self._scale = 1.0

    def bucket(self, bi: int = 0):
        """(layout, flat buffer, scale, present) of bucket `bi` after ``reduce(into_grads=False)``: what ``optim.Adam.step``
        takes; present[k] = parameter k had a gradient on this rank (the reference's optimiser skips the others -- e.g. the
        constructed-but-unused ``linear`` conv of standard_DCCRN, pvae_module.py:158 -- and so does the kernel)."""
        return self._tab[bi], self._flat[bi], self._scale, self._present[bi]

    def _reduce_device(self, bi: int, bucket, world: int, into_grads: bool):
        from . import optim
        tab = self._tab[bi]
        if tab is None:
            tab = self._tab[bi] = optim.TensorTable([p.numel() for p in bucket], bucket[0].device)
            self._flat[bi] = tab.flat()
        flat = self._flat[bi]
        grads = [None if p.grad is None else (p.grad if p.grad.is_contiguous() else p.grad.contiguous()) for p in bucket]
        self._present[bi] = [g is not None for g in grads]
        optim.bucket_gather(tab, grads, flat)
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)        # RCCL, on the device buffer
            self._scale = 1.0
        else:
            all_reduce_sum_(flat, self.group)
            self._scale = 1.0 / world
        if into_grads:
            # a parameter without a gradient keeps none (the reference's single-device optimiser skips it; see _check_presence)
            optim.bucket_scatter(tab, grads, flat, self._scale)
            for p, g in zip(bucket, grads):
                if g is not None and g is not p.grad:
                    p.grad = g                          # a non-contiguous .grad was averaged in its contiguous copy

    def _check_presence(self, bucket, world: int):
        """Once per bucket: every rank must agree on WHICH parameters have a gradient.  A parameter without one is skipped --
        it keeps ``.grad = None`` and the optimiser leaves it alone, as on the reference's single device (e.g. the
        constructed-but-unused ``linear`` conv of standard_DCCRN, pvae_module.py:158) -- which is only right if no other rank
        has a gradient for it.  One small all-reduce and one host read, on the first step only."""
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in bucket], dtype=torch.float32, device=bucket[0].device)
        all_reduce_sum_(flags, self.group)
        got = flags.cpu().tolist()
        bad = [k for k, v in enumerate(got) if v not in (0.0, float(world))]
        if bad:
            raise RuntimeError(f"GradAllReduce: parameters {bad[:8]} of a bucket have a gradient on some ranks only; the data-parallel "
                               "step expects the same graph on every rank")

    def reduce(self, into_grads: bool = True):
        """Average ``.grad`` over the ranks; a parameter without a gradient (on every rank: checked once) keeps none."""
        world = _world(self.group)
        if not _active(self.group):
            return
        for bi, bucket in enumerate(self.buckets):
            if not self._checked[bi]:
                self._check_presence(bucket, world)
                self._checked[bi] = True
            if bucket[0].is_cuda:
                self._reduce_device(bi, bucket, world, into_grads)
                continue
            n = sum(p.numel() for p in bucket)
            flat = self._flat[bi]
            if flat is None or flat.numel() != n or flat.device != bucket[0].device:
                flat = self._flat[bi] = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
            o = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[o:o + k].zero_()
                else:
                    flat[o:o + k].copy_(p.grad.reshape(-1))
                o += k
            all_reduce_sum_(flat, self.group)
            flat.div_(world)
            o = 0
            for p in bucket:
                k = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[o:o + k].view_as(p))
                o += k


def shard(t: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """This rank's contiguous slice of a global batch (equal shards keep the mean-of-shard-losses equal to the global mean)."""
    from .utils.dist_timing import shard_batch
    world = _world() if world is None else world
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    lo, hi = shard_batch(t.shape[0], rank, world)
    return t[lo:hi]
