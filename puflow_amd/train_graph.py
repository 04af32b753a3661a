"""The training step as hipGraphs.

An eager `TrainerModule.train_step` enqueues ~3 500 small kernels (one per autograd node of the un-fused training ops) and
is bound by host launch latency, not by the GPU.  The shapes of a step are static (fixed batch, fixed patch sizes), so the
whole step is captured once and replayed:

  single process : ONE graph = forward -> loss -> backward -> [one concatenation of the gradients] -> clip + Adam (two fused
                   launches with puflow_amd.optim.FusedClipAdam; PyTorch's capturable Adam costs ~550 per-tensor launches)
  world size > 1 : graph A = forward -> loss -> backward -> ONE concatenation of all gradients into a flat buffer
                   eager    = one RCCL all-reduce of the flat 806 103-float buffer, / world size
                   graph B = clip -> Adam   (every .grad is a view of the flat buffer by then)
The step starts from `.grad = None`: autograd installs the backward kernels' output tensors as the gradients (inside a
capture they live in the graph's pool, at fixed addresses), so there is neither a zero-fill nor an accumulate launch per
parameter.

Preconditions, all checked: the module is in train() mode with ActNorm initialised (the data-dependent first-batch init is a
host-side branch - run one eager `train_step` first, `GraphedTrainStep` does it for you when needed), SyncBN only on the fused
kernels over RCCL (their per-layer all-reduces of 257 doubles are captured; the row count travels with the sums), Adam built with `capturable=True` (`make_capturable` converts an
existing optimizer).  The learning rate lives in a device tensor: `set_lr()` (or a scheduler through `sync_lr()`) changes it
without re-capturing.  Results are those of the eager step with the same kernels in the same order; the NaN-loss guard of
the reference (train_pu1k.py:71-73) becomes a tensor select (trainer.training_step).
"""
from __future__ import annotations

from typing import Optional

import os
import torch
from torch import Tensor


# Capture in thread-local error mode: under the default (global) mode ANY thread's "unsafe" runtime call invalidates a running
# capture - and the RCCL process group's watchdog thread polls the events of earlier eager collectives (broadcast_module, the
# warm-up all-reduces) with hipEventQuery whenever it likes: "operation not permitted when stream is capturing", the watchdog
# dies and takes the process down.  Found by the one-rank RCCL smoke test, intermittently (a race with the watchdog's poll).
_CAPTURE_MODE = "thread_local"


def drain_collective_watchdog(dev) -> None:
    """Before a capture, under an RCCL process group: let the group's watchdog thread finish with the eager collectives issued
    so far (broadcasts, warm-up all-reduces, bench.py's start-up probe).  It polls their events every ~100 ms, and a poll that
    lands inside a capture has taken the process down ("operation not permitted when stream is capturing", intermittent: round 4
    moved the captures to thread-local error mode, round 5 saw it once more with one more eager collective in front).  The
    collectives are complete after the synchronize; one poll interval later the watchdog has dropped them."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
        import time
        torch.cuda.synchronize(dev)
        time.sleep(0.35)


def make_capturable(optimizer: torch.optim.Optimizer, device) -> torch.optim.Optimizer:
    """Adam(capturable=True) with the learning rate and the step counters as device tensors (required for capture)."""
    for g in optimizer.param_groups:
        g["capturable"] = True
        if not isinstance(g["lr"], Tensor):
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=device)
    for st in optimizer.state.values():
        if "step" in st and isinstance(st["step"], Tensor) and st["step"].device.type != "cuda":
            st["step"] = st["step"].to(device=device, dtype=torch.float32)
    return optimizer


# single-process captured step: the fused optimizer reads the gradients through a table of addresses instead of one concatenated
# buffer ("0": concatenate - the A/B reference)
_GRAD_TABLE = os.environ.get("PF_TRAIN_GRAD_TABLE", "1") != "0"


class GraphedTrainStep:
    def __init__(self, module, optimizer: torch.optim.Optimizer, batch, clip: float = 1e-2, warmup: int = 2):
        from . import train_ops
        from .dist import FlatGradBucket
        if getattr(getattr(module, "network", module), "sync_batchnorm", False) and train_ops._multi_rank():
            # SyncBN on the fused kernels all-reduces a layer's sums between two launches: RCCL collectives can be captured into
            # the graph, gloo's (host-side) cannot; the un-fused kernels read the global row count on the host
            if not train_ops._FUSED or torch.distributed.get_backend() != "nccl":
                raise RuntimeError("graphed_train_step with SyncBN needs the fused kernels and the nccl (RCCL) backend: its per-layer "
                                   "all-reduces are captured into the graph")
        self.module, self.optimizer, self.clip = module, optimizer, clip
        dev = next(module.parameters()).device
        from .dist import multi_rank
        self.world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                             torch.distributed.is_initialized()) else 1
        self.multi = multi_rank()           # the two-graph form with the eager all-reduce in between (dist.force_collectives: also with one rank)
        from .optim import FusedClipAdam
        self.fused = isinstance(optimizer, FusedClipAdam)
        self._gtab = self._gtab_zeros = None                # (the warm-up steps below run _update before the table exists)
        self._gtab_used = False
        if self.fused:
            optimizer.param_groups[0]["max_norm"] = clip
            optimizer.sync_lr()
        else:
            make_capturable(optimizer, dev)
        self.static = self._clone_batch(batch)
        module.train()
        if hasattr(module, "_nan_subs") and (module._nan_subs is None or module._nan_subs.device != dev):
            module._nan_subs = torch.zeros((), dtype=torch.int64, device=dev)     # NaN losses replaced inside the capture
        self.calls = 0
        self.bucket = FlatGradBucket(module.parameters())
        # warm-up AND capture run on this one stream: the fused kernels' zero-initialised scratch words (barrier words, the sticky
        # time-out word, statistics accumulators: train_ops._zeros_kept) are cached per (device, stream) and must be allocated
        # OUTSIDE the capture - inside it their torch.zeros would be a memset node that clears the sticky word on every replay
        side = self.capture_stream = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):               # ActNorm init, library loads, workspaces, Adam state
                module._sync_actnorm_init(self.static)
                self.warmup_loss = self._fwd_bwd()        # these ARE optimisation steps on `batch` (the fit loop counts them)
                self._reduce()
                self._update()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        drain_collective_watchdog(dev)
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_b: Optional[torch.cuda.CUDAGraph] = None
        if not self.multi and self.fused and _GRAD_TABLE:
            self._gtab = torch.zeros(len(optimizer.params), dtype=torch.int64, device=dev)
            self._gtab_zeros = torch.zeros(max(p.numel() for p in optimizer.params), dtype=torch.float32, device=dev)
        if not self.multi:
            with torch.cuda.graph(self.graph_a, stream=side, capture_error_mode=_CAPTURE_MODE):
                self.loss = self._fwd_bwd()
                self._update()
            if self._gtab_used:
                ptrs = optimizer.grad_table_of(self._gtab_zeros)
                if ptrs is None:
                    raise RuntimeError("graphed_train_step: a captured gradient is not a contiguous fp32 tensor of its parameter's "
                                       "size (set PF_TRAIN_GRAD_TABLE=0 to concatenate the gradients instead)")
                self._gtab.copy_(torch.tensor(ptrs, dtype=torch.int64))
                self._gtab_keep = [p.grad for p in optimizer.params]          # the captured tensors: their memory IS the table's targets
                torch.cuda.synchronize(dev)
        else:
            with torch.cuda.graph(self.graph_a, stream=side, capture_error_mode=_CAPTURE_MODE):
                self.loss = self._fwd_bwd()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=side, capture_error_mode=_CAPTURE_MODE):
                self._update()

    # ---- pieces (the same calls in warm-up, capture and - implicitly - replay)
    @staticmethod
    def _clone_batch(batch):
        if isinstance(batch, dict):
            return {k: (v.clone() if isinstance(v, Tensor) else v) for k, v in batch.items()}
        return tuple(v.clone() if isinstance(v, Tensor) else v for v in batch)

    def _copy_in(self, batch) -> None:
        """The batch into the captured step's input tensors: ONE multi-tensor launch for the tensors that can take it (same
        dtype and device as their destination), one copy each for the rest (it was one launch per tensor of the batch,
        six per step, each with its own gap in front of the replay)."""
        if isinstance(batch, dict):
            pairs = [(self.static[k], v) for k, v in batch.items() if isinstance(v, Tensor)]
        else:
            pairs = [(d, v) for d, v in zip(self.static, batch) if isinstance(v, Tensor)]
        pairs = [(d, v) for d, v in pairs if d.data_ptr() != v.data_ptr() or d.shape != v.shape]      # already in place
        fast = [(d, v) for d, v in pairs if v.dtype == d.dtype and v.device == d.device and v.shape == d.shape]
        words = [(d, v) for d, v in fast if d.is_cuda and d.element_size() == 4 and d.is_contiguous() and v.is_contiguous()
                 and d.numel() > 0]
        if len(words) > 1:                                  # 32-bit tensors on the device: one launch of ours for up to 8 of them
            import ctypes
            from . import _lib
            from .train_ops import _stream
            lib = _lib.load()
            for k in range(0, len(words), 8):
                grp = words[k:k + 8]
                n = len(grp)
                src = (ctypes.c_void_p * n)(*[v.data_ptr() for _, v in grp])
                dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in grp])
                cnt = (ctypes.c_longlong * n)(*[d.numel() for d, _ in grp])
                _lib.check(lib.pf_copy_n(src, dst, cnt, n, _stream()), "pf_copy_n")
            fast = words
        elif len(fast) > 1:
            torch._foreach_copy_([d for d, _ in fast], [v for _, v in fast])
        else:
            fast = []
        for d, v in pairs:
            if not any(d is f for f, _ in fast):
                d.copy_(v)

    def _fwd_bwd(self) -> Tensor:
        self.bucket.drop_grads()
        loss = self.module.training_step(self.static, 0)
        loss.backward(self.module.backward_seed(loss) if hasattr(self.module, "backward_seed") else None)
        if self.multi:
            self.bucket.pack()
        return loss.detach()

    def _reduce(self) -> None:
        if self.multi:
            torch.distributed.all_reduce(self.bucket.flat)
            self.bucket.flat.div_(self.world)

    def _update(self) -> None:
        if self.fused:                                      # clip + Adam as two launches on the flat gradient buffer
            if self._gtab is not None and torch.cuda.is_current_stream_capturing():
                # single process, inside the capture: the optimizer reads the gradients where the backward left them, through a
                # table of their addresses that __init__ fills right after the capture (they are the same at every replay) -
                # no concatenation launches in front of the update
                self._gtab_used = True
                self.optimizer.step_table(self._gtab)
                return
            if not self.multi:
                self.bucket.pack()
            self.optimizer.step_flat(self.bucket.flat)
            return
        torch.nn.utils.clip_grad_norm_(self.bucket.params, self.clip, foreach=True)
        self.optimizer.step()

    # ---- public
    def __call__(self, batch) -> Tensor:
        if self.fused:
            self.optimizer.sync_lr()                      # a scheduler may have edited param_groups[0]["lr"] since the last call
        self._copy_in(batch)
        self.graph_a.replay()
        if self.graph_b is not None:
            self._reduce()
            self.graph_b.replay()
        # a replay updates parameters and BatchNorm running statistics without passing through torch's version counters:
        # bump them, or a later eval forward would keep a packed plan of the OLD weights (interpflow._signature)
        if getattr(self, "_written", None) is None:
            self._written = list(self.module.parameters()) + list(self.module.buffers())
        torch._C._increment_version(self._written)
        self.calls += 1
        if self.calls % self.CHECK_EVERY == 0:
            self.check()
        return self.loss

    CHECK_EVERY = 64          # replays between two reads of the device status words (each read is a synchronisation)

    def check(self) -> int:
        """A timed-out EMD barrier raises; returns the NaN losses replaced since the last check (trainer.check_device_status)."""
        fn = getattr(self.module, "check_device_status", None)
        return fn() if fn is not None else 0

    def set_lr(self, lr: float) -> None:
        if self.fused:
            self.optimizer.param_groups[0]["lr"] = float(lr)
            self.optimizer.sync_lr()                        # the device scalar the captured kernels read
            return
        for g in self.optimizer.param_groups:
            if isinstance(g["lr"], Tensor):
                g["lr"].fill_(float(lr))
            else:                                           # a scheduler replaced the tensor by a float: restore the tensor
                g["lr"] = torch.tensor(float(lr), dtype=torch.float32, device=self.bucket.flat.device)
                raise RuntimeError("the captured learning-rate tensor was replaced; use sync_lr() after scheduler.step()")

    def sync_lr(self, scheduler_lr_holder) -> None:
        """After `scheduler.step(metric)` on a shadow optimizer / param-group list holding plain floats."""
        if self.fused:
            self.set_lr(float(scheduler_lr_holder[0]["lr"]))
            return
        for g, h in zip(self.optimizer.param_groups, scheduler_lr_holder):
            g["lr"].fill_(float(h["lr"]))
