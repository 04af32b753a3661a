"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

The path shards naturally: patches are independent, so inference splits the patch
batch contiguously over ranks with NO collective.  The training step adds exactly one collective:
an all-reduce of the flat gradient (806 103 fp32 = 3.2 MB).  At that size the ring is latency-bound on
xGMI, so the gradients travel as ONE bucket, one call per step; nothing is overlapped.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` patches owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def gather_shards(x_local: torch.Tensor, total: int, rank: int, world: int) -> torch.Tensor:
    """All ranks' output shards back in batch order (only needed when a merged cloud is wanted)."""
    if world == 1:
        return x_local
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]] = x_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


class FlatGradBucket:
    """All gradients of a module as ONE flat fp32 buffer -> a single all-reduce per step.

    as_views=True (what the trainer uses): every parameter's `.grad` IS a view of the flat buffer, autograd accumulates
    into it in place, `flat.zero_()` replaces `optimizer.zero_grad()`, and the all-reduce needs no packing copies at all.
    (Do not call `zero_grad(set_to_none=True)` on such parameters: it would drop the views; `rebind()` restores them.)
    as_views=False keeps the round-1 behaviour: gradients are copied in and out around the collective."""

    def __init__(self, params: Iterable[torch.nn.Parameter], as_views: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.as_views = as_views
        if as_views:
            self.rebind()

    def rebind(self) -> None:
        o = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[o:o + n].view_as(p)
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
            o += n

    def views_intact(self) -> bool:
        o = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                return False
            o += p.numel()
        return True

    def all_reduce_mean(self) -> None:
        """grad <- mean over ranks of grad (missing grads count as zero)."""
        multi = dist.is_initialized() and dist.get_world_size() > 1
        if self.as_views:
            if not self.views_intact():
                self.rebind()
            if multi:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.flat.div_(dist.get_world_size())
            return
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[o:o + n].zero_()
            else:
                self.flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        if multi:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        o = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[o:o + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += n


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Same weights / buffers on every rank (e.g. after ActNorm's data-dependent init on rank 0)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def broadcast_buffers(module: torch.nn.Module, src: int = 0) -> None:
    """BatchNorm running statistics are updated from each rank's own shard (local batch statistics, like
    DistributedDataParallel without SyncBN); like DDP's `broadcast_buffers`, rank `src`'s copies are made
    authoritative before anything reads them in eval mode (validation, checkpoints)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in module.buffers():
        dist.broadcast(t.data, src=src)


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
