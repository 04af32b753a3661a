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


_FORCE_COLLECTIVES = False


def force_collectives(on: bool = True) -> None:
    """Run the multi-rank code path (gradient bucket all-reduce between the two step graphs, module / buffer broadcasts,
    ActNorm-init broadcast, SyncBN statistics all-reduce) even in a process group of ONE rank.  The arithmetic is unchanged
    (a one-rank all-reduce is the identity, the mean divides by 1); it exists so that the RCCL backend, the interleaving of
    eager collectives with graph replays and the group's teardown can be exercised on a single GPU
    (tests/test_gpu_rccl_smoke.py, `bench.py` with PF_BENCH_FORCE_DIST=1)."""
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


def multi_rank() -> bool:
    """True when the collectives of the training path have to run: an initialised process group of more than one rank (or of
    one rank under `force_collectives`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _FORCE_COLLECTIVES


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

    as_views=False (what the trainer uses): the step starts with `.grad = None` for every parameter, so autograd hands the
    gradient tensors of the backward kernels over as they are (no zero-fill, no accumulate kernel per parameter - at
    ~230 parameter tensors those were ~230 launches per step, more than the collective costs).  `all_reduce_mean()` packs
    them with ONE concatenation into the flat buffer, all-reduces it, and re-points every `.grad` at its slice of the
    buffer (views: no copy back), which is what clipping and Adam then read.
    as_views=True: every `.grad` IS a view of the flat buffer from the start and autograd accumulates in place (zero the
    buffer with `flat.zero_()`; do not call `zero_grad(set_to_none=True)`, `rebind()` restores the views): no packing
    copy at all, but one accumulate launch per parameter."""

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

    def drop_grads(self) -> None:
        """Start of a step in the packing mode: autograd will install fresh gradient tensors."""
        for p in self.params:
            p.grad = None

    def pack(self) -> None:
        """flat <- concatenation of all gradients (missing ones count as zero), ONE launch; `.grad` <- views of flat."""
        torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params], out=self.flat)
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n

    def all_reduce_mean(self, always_pack: bool = False) -> None:
        """grad <- mean over ranks of grad (missing grads count as zero).  always_pack: concatenate into the flat buffer in a
        single process too (the fused optimizer reads the flat buffer)."""
        multi = multi_rank()
        if self.as_views:
            if not self.views_intact():
                self.rebind()
        else:
            if not multi and not always_pack:
                return                                  # single process: the gradients stay where autograd put them
            self.pack()
        if multi:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def _broadcast_tensors(tensors, src: int) -> None:
    """One broadcast per dtype (the tensors of a dtype packed into a flat buffer) instead of one per tensor; the values are
    written back with in-place copies under no_grad, which bump the tensors' version counters - a packed eval plan keyed on
    them (interpflow._signature) therefore sees the new values (writes through `.data` would not)."""
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for ts in groups.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src)
            pos = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[pos:pos + n].view_as(t))
                pos += n


def broadcast_module(module: torch.nn.Module, src: int = 0) -> None:
    """Same weights / buffers on every rank (e.g. after ActNorm's data-dependent init on rank 0)."""
    if not multi_rank():
        return
    _broadcast_tensors(list(module.parameters()) + list(module.buffers()), src)


def broadcast_buffers(module: torch.nn.Module, src: int = 0) -> None:
    """BatchNorm running statistics are updated from each rank's own shard (local batch statistics, like
    DistributedDataParallel without SyncBN); like DDP's `broadcast_buffers`, rank `src`'s copies are made
    authoritative before anything reads them in eval mode (validation, checkpoints)."""
    if not multi_rank():
        return
    _broadcast_tensors(list(module.buffers()), src)


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
