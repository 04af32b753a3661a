"""autograd wiring of the HIP training kernels (csrc/train_ops.hip) and the train-mode forward of
PointInterpFlow composed from them.

Split between HIP and torch in TRAINING mode (documented in DESIGN.md section 7):
  HIP  : every Conv2d(1x1)/Linear (forward, dX, dW, db), BatchNorm(train)+LeakyReLU, ReLU/LeakyReLU,
         edge-feature gather / scatter-add, max-pool over K, neighbour gather backward, softmax-weighted
         latent sum, repeat_interleave backward, kNN, Chamfer, EMD.
         ActNorm, coupling + reverse + injector (both directions), per-batch log-det / Gaussian sums.
  torch: tensor re-layout (cat / slice / reshape / index and their autograd), parameter-only scalars
         (sum(logs), slogdet / inverse of the 3x3 W), [B]-sized loss bookkeeping, and the optimiser
         (torch.optim.Adam, exactly as in the reference).
Activations are channels-last [rows, C] fp32 (rows = points or edges).
"""
from __future__ import annotations

import ctypes
import os
import sys

import math
from typing import List, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib

LOG2PI = float(math.log(2 * math.pi))
_WS = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ws(dev, n: int) -> Tensor:
    """Grow-only fp32 scratch per (device, stream): reuse is stream-ordered - every kernel that uses it is enqueued
    on that stream before the next one overwrites it - so two streams never share a buffer."""
    key = (dev, _stream())
    t = _WS.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=dev)
        _WS[key] = t
    return t


# Matrix-pipe arithmetic of the training GEMMs: "f32" (default) = every GEMM on the f32 MFMA (bit-exact fp32 fma chains);
# "split" = forward GEMMs as split-fp16 products (3 fp16 MFMAs per 32-deep step), GEMMs with a gradient operand as
# split-bf16 (6 bf16 MFMAs, fp32 exponent range) - csrc/train_ops.hip gemm_split_kernel.  Measured at 32 x (256 -> 1024):
# the same step time (the layer GEMMs of this un-fused path are bound by staging and launch count, not by the MFMA rate:
# profiles/r2_train), so the exact arithmetic stays the default; the gradient tests pass in both modes.
_GEMM_MODE = os.environ.get("PF_TRAIN_GEMM", "f32")
ARITH_FWD, ARITH_BWD = (2, 3) if _GEMM_MODE == "split" else (0, 0)


_STAT = {}


def _zeros_kept(n: int, dtype, dev) -> Tensor:
    """Zero-initialised device words that the kernels keep zero themselves (barrier words, statistics accumulators) or that are
    STICKY across launches (the time-out word).  They must exist before a hipGraph capture starts: a `torch.zeros` inside a
    capture becomes a memset node that clears them on every replay - a time-out recorded by replay i would be erased by replay
    i + 1 before the host ever looked.  GraphedTrainStep warms up on its capture stream, so the (device, stream) entries are
    there; anything else that captures these kernels must run them once eagerly on the capture stream first."""
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("puflow_amd.train_ops: the zero-initialised scratch words of the fused training kernels would be allocated "
                           "inside a hipGraph capture (a memset node would clear the sticky status word on every replay): run the "
                           "step once eagerly on the capture stream before capturing it (train_graph.GraphedTrainStep does)")
    return torch.zeros(n, dtype=dtype, device=dev)


def _stat(dev) -> Tensor:
    """4097 doubles (PF_TRAIN_STAT_DOUBLES) per (device, stream): the column-statistics accumulators of the fused training kernels."""
    key = (dev, _stream())
    t = _STAT.get(key)
    if t is None:
        t = _zeros_kept(4097, torch.float64, dev)
        _STAT[key] = t
    return t


_SYNCW = {}
# persistent (grid-barrier) kernels of the training step (csrc/train_fused.hip: ec_fwdp_kernel): "1" = the main chain's EdgeConv
# units may use them (the default; the network attribute `train_persistent` and a device shared between processes switch them
# off per module), "0" = the per-layer kernels everywhere (the A/B reference).  Read once at import.
_PERSIST = os.environ.get("PF_TRAIN_PERSIST", "1") != "0"


def _sync_words(dev) -> Tensor:
    """4 zero-initialised 32-bit words per (device, stream): arrivals / generation / exits of the persistent kernels' grid
    barriers (left zero by every launch) and a sticky status word (check_persist_status)."""
    key = (dev, _stream())
    t = _SYNCW.get(key)
    if t is None:
        t = _zeros_kept(4, torch.int32, dev)
        _SYNCW[key] = t
    return t


def check_persist_status(device=None) -> None:
    """Raise if a grid barrier of a persistent training kernel timed out since the last check (its workgroups were not all
    resident: another barrier kernel or another process held the CUs).  The unit's output was NaN, the optimizer skipped the
    update on the device; the statistics accumulators of that stream are cleared here.  One word read per stream that used a
    persistent kernel: call where the host synchronises anyway."""
    for (dev, stream), t in list(_SYNCW.items()):
        if device is not None and torch.device(device) != dev:
            continue
        if int(t[3].item()):
            t.zero_()
            st = _STAT.get((dev, stream))
            if st is not None:
                st.zero_()
            raise _lib.PuflowHipError("a persistent training kernel timed out on a grid barrier (workgroups not co-resident - is the GPU "
                                      "shared with another process, or did two barrier kernels run side by side?).  Set "
                                      "net.train_persistent = False (cfg.persistent_kernels = False) on a shared device")


# ---- SyncBN on the fused kernels: the library leaves a BatchNorm layer's LOCAL column sums in `sync_sums` and calls back;
# the callback all-reduces them (stream-ordered on torch's current stream, which is the stream the kernels were enqueued on)
_SYNC_SUMS = {}
_SYNC_CB_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)


def _sync_sums(dev) -> Tensor:
    key = (dev, _stream())
    t = _SYNC_SUMS.get(key)
    if t is None:
        t = _SYNC_SUMS[key] = _zeros_kept(2 * 128 + 1, torch.float64, dev)
    return t


def _sync_cb_impl(user, sums, n, stream):
    try:
        import torch.distributed as dist
        for t in _SYNC_SUMS.values():
            if t.data_ptr() == sums:
                dist.all_reduce(t)                      # SUM over the ranks: 2 x 128 column sums + the row count
                return 0
        return 1
    except Exception as ex:                             # an exception must not unwind through the C caller
        print(f"puflow_amd: SyncBN all-reduce failed: {type(ex).__name__}: {ex}", file=sys.stderr)
        return 2


_SYNC_CB = _SYNC_CB_T(_sync_cb_impl)                    # module-level: must outlive every call


def _attach_sync(d, dev) -> None:
    """Global-batch BatchNorm statistics for a fused-kernel call (PfEcTrain / PfBnMlpTrain)."""
    d.sync_cb = ctypes.cast(_SYNC_CB, ctypes.c_void_p)
    d.sync_user = None
    d.sync_sums = _sync_sums(dev).data_ptr()


def _gemm(A: Tensor, sam: int, sak: int, Bm: Tensor, sbk: int, sbn: int, C: Tensor, ldc: int, bias, M: int, N: int, K: int,
          arith: int = 0):
    lib = _lib.load()
    need = lib.pf_gemm_ws_floats(M, N, K)
    ws = _ws(C.device, need) if need else None
    _lib.check(lib.pf_gemm_ex(arith, A.data_ptr(), sam, sak, Bm.data_ptr(), sbk, sbn, C.data_ptr(), ldc,
                              bias.data_ptr() if bias is not None else None, M, N, K,
                              ws.data_ptr() if ws is not None else None, need, _stream()), "pf_gemm")


class LinearFn(Function):
    """y[R,Cout] = x[R,Cin] W[Cout,Cin]^T + b   (nn.Linear / Conv2d 1x1 on channels-last rows)."""

    @staticmethod
    def forward(ctx, x, W, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        W = W.contiguous()
        R, Cin = x2.shape
        Cout = W.shape[0]
        y = torch.empty((R, Cout), dtype=torch.float32, device=x.device)
        _gemm(x2, Cin, 1, W, 1, Cin, y, Cout, b, R, Cout, Cin, ARITH_FWD)
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.shp = shp
        return y.view(*shp[:-1], Cout)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        R, Cin = x2.shape
        Cout = W.shape[0]
        dy2 = dy.reshape(R, Cout).contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            _gemm(dy2, Cout, 1, W, Cin, 1, dx, Cin, None, R, Cin, Cout, ARITH_BWD)
            dx = dx.view(ctx.shp)
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            _gemm(dy2, 1, Cout, x2, Cin, 1, dW, Cin, None, Cout, Cin, R, ARITH_BWD)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            lib = _lib.load()
            db = torch.empty((Cout,), dtype=torch.float32, device=dy.device)
            ws = _ws(dy.device, 2 * lib.pf_bn_chunks(R) * Cout)
            _lib.check(lib.pf_colsum(dy2.data_ptr(), R, Cout, db.data_ptr(), ws.data_ptr(), _stream()), "pf_colsum")
        return dx, dW, db


def linear(x: Tensor, W: Tensor, b=None) -> Tensor:
    return LinearFn.apply(x, W.reshape(W.shape[0], -1), b)


class BnLreluFn(Function):
    """BatchNorm(training, batch statistics over rows) + LeakyReLU; running stats updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, slope, eps, momentum):
        lib = _lib.load()
        x = x.contiguous()
        R, C = x.shape
        y = torch.empty_like(x)
        save = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws = _ws(x.device, (2 * lib.pf_bn_chunks(R) + 2) * C)
        g, b = gamma.contiguous(), beta.contiguous()
        _lib.check(lib.pf_bn_lrelu_fwd(x.data_ptr(), R, C, g.data_ptr(), b.data_ptr(), slope, eps, momentum,
                                       run_mean.data_ptr() if run_mean is not None else None,
                                       run_var.data_ptr() if run_var is not None else None, y.data_ptr(), save.data_ptr(),
                                       ws.data_ptr(), _stream()), "pf_bn_lrelu_fwd")
        ctx.save_for_backward(x, g, b, save)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, g, b, save = ctx.saved_tensors
        R, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty((C,), dtype=torch.float32, device=x.device)
        db = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = _ws(x.device, (2 * lib.pf_bn_chunks(R) + 2) * C)
        _lib.check(lib.pf_bn_lrelu_bwd(x.data_ptr(), dy.data_ptr(), R, C, g.data_ptr(), b.data_ptr(), ctx.slope,
                                       save.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                       _stream()), "pf_bn_lrelu_bwd")
        return dx, dg, db, None, None, None, None, None


class SyncBnLreluFn(Function):
    """BnLreluFn with statistics over the GLOBAL batch (all ranks): the per-column sums are all-reduced between the
    kernel stages - 2 small all-reduces forward (mean, then centred variance: the same two-pass scheme as the local
    kernel), 1 backward.  dgamma / dbeta stay the LOCAL sums (the gradient bucket's mean over ranks then gives the
    gradient of the averaged loss, as with torch.nn.SyncBatchNorm under DDP); dx uses the global means."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, slope, eps, momentum):
        import torch.distributed as dist
        lib = _lib.load()
        x = x.contiguous()
        R, C = x.shape
        dev = x.device
        ws = _ws(dev, 2 * lib.pf_bn_chunks(R) * C)
        g, b = gamma.contiguous(), beta.contiguous()
        stat = torch.empty((C + 1,), dtype=torch.float32, device=dev)
        _lib.check(lib.pf_bn_colstat(x.data_ptr(), R, C, None, stat.data_ptr(), ws.data_ptr(), _stream()), "pf_bn_colstat")
        stat[C] = float(R)
        dist.all_reduce(stat)
        Rg = float(stat[C].item())
        mean = (stat[:C] / Rg).contiguous()
        var = torch.empty((C,), dtype=torch.float32, device=dev)
        _lib.check(lib.pf_bn_colstat(x.data_ptr(), R, C, mean.data_ptr(), var.data_ptr(), ws.data_ptr(), _stream()), "pf_bn_colstat")
        dist.all_reduce(var)
        var = (var / Rg).contiguous()
        y = torch.empty_like(x)
        save = torch.empty((2, C), dtype=torch.float32, device=dev)
        _lib.check(lib.pf_bn_apply_stats(x.data_ptr(), R, C, mean.data_ptr(), var.data_ptr(), Rg / max(Rg - 1.0, 1.0),
                                         g.data_ptr(), b.data_ptr(), slope, eps, momentum,
                                         run_mean.data_ptr() if run_mean is not None else None,
                                         run_var.data_ptr() if run_var is not None else None, y.data_ptr(), save.data_ptr(),
                                         _stream()), "pf_bn_apply_stats")
        ctx.save_for_backward(x, g, b, save)
        ctx.slope, ctx.Rg = slope, Rg
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        lib = _lib.load()
        x, g, b, save = ctx.saved_tensors
        R, C = x.shape
        dy = dy.contiguous()
        ws = _ws(x.device, 2 * lib.pf_bn_chunks(R) * C)
        sums = torch.empty((2, C), dtype=torch.float32, device=x.device)
        _lib.check(lib.pf_bn_bwd_sums(x.data_ptr(), dy.data_ptr(), R, C, g.data_ptr(), b.data_ptr(), ctx.slope, save.data_ptr(),
                                      sums.data_ptr(), ws.data_ptr(), _stream()), "pf_bn_bwd_sums")
        db, dg = sums[0].clone(), sums[1].clone()
        dist.all_reduce(sums)
        means = (sums / ctx.Rg).contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.pf_bn_bwd_apply(x.data_ptr(), dy.data_ptr(), R, C, g.data_ptr(), b.data_ptr(), ctx.slope, save.data_ptr(),
                                       means.data_ptr(), dx.data_ptr(), _stream()), "pf_bn_bwd_apply")
        return dx, dg, db, None, None, None, None, None


# BatchNorm statistics over all ranks: an ARGUMENT of the forward, not process state - `forward_train` takes it from the
# module (`PointInterpFlow.sync_batchnorm`, set from `cfg.sync_batchnorm`) and holds it in this context variable for the
# duration of that call; `sync_bn(True)` is the same scope for callers of the single ops (tests)
import contextlib
import contextvars

_SYNC_BN: contextvars.ContextVar = contextvars.ContextVar("puflow_sync_bn", default=False)


@contextlib.contextmanager
def sync_bn(on: bool):
    tok = _SYNC_BN.set(bool(on))
    try:
        yield
    finally:
        _SYNC_BN.reset(tok)


# Bit-reproducible training steps (debugging switch; VERDICT / ADVICE r4): `net.deterministic = True` (TrainerModule:
# `cfg.deterministic`).  forward_train copies it here at the start of every train-mode forward and the autograd functions read it
# when they run (the backward of that forward included): BatchNorm statistics as exact 64-bit fixed-point sums
# (PF_TRAIN_DETERMINISTIC: csrc/train_fused.hip stat_add; the persistent kernels are not used), the latent's gradient as a gather
# over the sorted transposed neighbour lists, the Chamfer gradient by pf_chamfer_bwd_det.  Everything else in the fused step is
# already order-fixed (split-K partials reduced in index order, the dQ gather over the now sorted lists, the auction).
_DET = False


def set_deterministic(on: bool) -> None:
    global _DET
    _DET = bool(on)


def deterministic() -> bool:
    return _DET


# Weight gradients beside the backward chain (csrc/api.hip pf_train_set_dw_stream) - an opt-in that did NOT pay at the bench
# shape.  The backward of a unit produces the gradient of its input - which the unit before it waits for - and the gradients of
# its weights, which nothing reads before the optimizer: with the switch on (`net.train_dw_stream = True` / cfg.dw_stream, or
# PF_TRAIN_DW_STREAM=1; =0 forces it off) the split-K weight-gradient kernels and their reductions of the EdgeConv units and the
# conditioner / merge MLPs go to one more stream (a third parallel branch of the captured step), ~0.9 ms of the main chain's
# ~3.9 ms of kernels at 32 x (256 -> 1024).  Their workspaces come from that stream's own pool, every buffer they read is kept
# alive until the join, and the join - the calling stream waits for the weight-gradient stream - is an autograd end-of-pass
# callback, queued by the first backward function that uses the stream: whoever reads `.grad` after `backward()` sees finished
# gradients, eager or captured.  Same kernels, same arithmetic, same bits (tests/test_gpu_train.py).  Measured, same box, two
# rounds each: 4.54 -> 4.79 ms per captured step with the persistent EdgeConv kernels, 5.22 -> 5.31 without - the step is bound
# by the SUM of its kernels' work (each of them fills the chip), not by the length of the dependent chain, and kernels that share
# the chip slow each other down by more than the chain gets shorter.  (The flow chains stay on the calling stream in any case:
# f and g share their parameters, so autograd adds their gradients there before any join.)
_DW_ENV = os.environ.get("PF_TRAIN_DW_STREAM")
_DW_NET = False
_DW_STREAMS = {}
_DW_PASS = {"pending": False, "keep": [], "streams": []}


def _dw_join() -> None:
    """End of the autograd pass (calling thread): the consumer of the gradients waits for the weight-gradient stream."""
    for st in _DW_PASS["streams"]:
        torch.cuda.current_stream(st.device).wait_stream(st)
    _DW_PASS["pending"] = False
    _DW_PASS["keep"] = []
    _DW_PASS["streams"] = []


def _dw_begin(dev, *keep):
    """The weight-gradient stream for one backward call on `dev`, or None (switch off, SyncBN, not inside an autograd pass).
    `keep`: tensors the side kernels read - held until the join so that the allocator cannot hand their memory to the main chain."""
    if not (_DW_ENV == "1" or (_DW_ENV is None and _DW_NET)) or _sync_bn_active():
        return None
    # one join per autograd pass, keyed by the pass's id: a pass that died with an exception must not leave the next one without
    tid = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else 0
    if tid < 0:
        return None                                        # a backward function called by hand, outside an autograd pass
    if _DW_PASS["pending"] is not True or _DW_PASS.get("task") != tid:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_dw_join)
        except RuntimeError:
            return None
        if _DW_PASS.get("task") != tid:                    # leftovers of a pass that never reached its join
            _DW_PASS["keep"], _DW_PASS["streams"] = [], []
        _DW_PASS["pending"], _DW_PASS["task"] = True, tid
    st = _DW_STREAMS.get(dev)
    if st is None:
        st = _DW_STREAMS[dev] = torch.cuda.Stream(device=dev)
    if st not in _DW_PASS["streams"]:
        _DW_PASS["streams"].append(st)
    _DW_PASS["keep"].append(keep)
    return st


def _dw_ws(st, dev, n: int) -> Tensor:
    with torch.cuda.stream(st):
        return _ws(dev, n)


class _dw_call:
    """`with _dw_call(st):` around ONE backward entry point: the library's weight-gradient stream is a per-thread setting (the
    autograd engine runs backward functions on its own device threads), set for exactly that call."""

    def __init__(self, st):
        self.st = st

    def __enter__(self):
        if self.st is not None:
            _lib.load().pf_train_set_dw_stream(self.st.cuda_stream)

    def __exit__(self, *exc):
        if self.st is not None:
            _lib.load().pf_train_set_dw_stream(None)
        return False


def _multi_rank() -> bool:
    import torch.distributed as dist
    from .dist import multi_rank
    return multi_rank()


def _sync_bn_active() -> bool:
    return bool(_SYNC_BN.get()) and _multi_rank()


def bn_lrelu(x: Tensor, bn: torch.nn.BatchNorm2d, slope: float) -> Tensor:
    fn = SyncBnLreluFn if _sync_bn_active() else BnLreluFn
    y = fn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, slope, bn.eps, bn.momentum)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    return y


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(lib.pf_act_fwd(x.data_ptr(), slope, x.numel(), y.data_ptr(), _stream()), "pf_act_fwd")
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(lib.pf_act_bwd(y.data_ptr(), dy.data_ptr(), ctx.slope, y.numel(), dx.data_ptr(), _stream()), "pf_act_bwd")
        return dx, None


class EdgeFeatureFn(Function):
    """x [B,N,C], idx int32 [B,N,K] -> [B*N*K, 3C] = [x_i, x_j, x_j - x_i]."""

    @staticmethod
    def forward(ctx, x, idx):
        lib = _lib.load()
        x = x.contiguous()
        B, N, C = x.shape
        K = idx.shape[-1]
        out = torch.empty((B * N * K, 3 * C), dtype=torch.float32, device=x.device)
        _lib.check(lib.pf_edge_feature_fwd(x.data_ptr(), idx.data_ptr(), B, N, K, C, out.data_ptr(), _stream()), "pf_edge_feature_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, K, C)
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        B, N, K, C = ctx.dims
        g = g.contiguous()
        dx = torch.zeros((B, N, C), dtype=torch.float32, device=g.device)
        _lib.check(lib.pf_edge_feature_bwd(g.data_ptr(), idx.data_ptr(), B, N, K, C, dx.data_ptr(), _stream()), "pf_edge_feature_bwd")
        return dx, None


class MaxPoolKFn(Function):
    """y [T*K, C] -> max over the K rows of each point [T, C]."""

    @staticmethod
    def forward(ctx, y, K):
        lib = _lib.load()
        y = y.contiguous()
        C = y.shape[1]
        T = y.shape[0] // K
        out = torch.empty((T, C), dtype=torch.float32, device=y.device)
        arg = torch.empty((T, C), dtype=torch.int32, device=y.device)
        _lib.check(lib.pf_maxpool_k_fwd(y.data_ptr(), T, K, C, out.data_ptr(), arg.data_ptr(), _stream()), "pf_maxpool_k_fwd")
        ctx.save_for_backward(arg)
        ctx.dims = (T, K, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        T, K, C = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((T * K, C), dtype=torch.float32, device=dy.device)
        _lib.check(lib.pf_maxpool_k_bwd(dy.data_ptr(), arg.data_ptr(), T, K, C, dx.data_ptr(), _stream()), "pf_maxpool_k_bwd")
        return dx, None


class GatherRowsFn(Function):
    """z [B,N,C], idx int32 [B,N,K] -> z[b, idx] as [B*N*K, C] (forward = indexing, backward = HIP scatter-add)."""

    @staticmethod
    def forward(ctx, z, idx):
        B, N, C = z.shape
        K = idx.shape[-1]
        out = z[torch.arange(B, device=z.device).view(B, 1, 1), idx.long()].reshape(B * N * K, C)
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, K, C)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        B, N, K, C = ctx.dims
        g = g.contiguous()
        if _DET:                                              # ordered gather over the sorted transposed lists: no float atomics
            off, edge = knn_csr(idx.contiguous())
            dz = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
            _lib.check(lib.pf_scatter_rows_det(g.data_ptr(), off.data_ptr(), edge.data_ptr(), B * N, C, dz.data_ptr(), _stream()),
                       "pf_scatter_rows_det")
            return dz, None
        dz = torch.zeros((B, N, C), dtype=torch.float32, device=g.device)
        _lib.check(lib.pf_scatter_rows(g.data_ptr(), idx.data_ptr(), B, N, K, C, dz.data_ptr(), _stream()), "pf_scatter_rows")
        return dz, None


class RepeatRowsFn(Function):
    """repeat_interleave(c, R, dim=1): forward = data movement, backward = HIP group sum."""

    @staticmethod
    def forward(ctx, c, R):
        ctx.R = R
        ctx.shp = c.shape
        return torch.repeat_interleave(c, R, dim=1)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, N, C = ctx.shp
        g = g.contiguous()
        out = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
        _lib.check(lib.pf_group_sum(g.data_ptr(), B * N, ctx.R, C, out.data_ptr(), _stream()), "pf_group_sum")
        return out, None


class SoftmaxWsumFn(Function):
    """w [T,K,ldw] logits (first R channels used), zj [T,K,3] -> fz [T,3,R]."""

    @staticmethod
    def forward(ctx, w, zj, R):
        lib = _lib.load()
        w, zj = w.contiguous(), zj.contiguous()
        T, K, ldw = w.shape
        a = torch.empty((T, K, R), dtype=torch.float32, device=w.device)
        fz = torch.empty((T, 3, R), dtype=torch.float32, device=w.device)
        _lib.check(lib.pf_softmax_wsum_fwd(w.data_ptr(), ldw, zj.data_ptr(), K, R, T, a.data_ptr(), fz.data_ptr(), _stream()),
                   "pf_softmax_wsum_fwd")
        ctx.save_for_backward(a, zj)
        ctx.dims = (T, K, R, ldw)
        return fz

    @staticmethod
    def backward(ctx, dfz):
        lib = _lib.load()
        a, zj = ctx.saved_tensors
        T, K, R, ldw = ctx.dims
        dfz = dfz.contiguous()
        dw = torch.empty((T, K, ldw), dtype=torch.float32, device=dfz.device)
        dzj = torch.empty((T, K, 3), dtype=torch.float32, device=dfz.device)
        _lib.check(lib.pf_softmax_wsum_bwd(a.data_ptr(), zj.data_ptr(), dfz.data_ptr(), K, R, ldw, T, dw.data_ptr(),
                                           dzj.data_ptr(), _stream()), "pf_softmax_wsum_bwd")
        return dw, dzj, None


class InterpWsumFn(Function):
    """Interpolation of the latent (interpflow.py:153-186, 312-318): w [T,8,ldw] logits (first R channels), z [B,N,3], idx8 int32
    [B,N,16|8] -> u [B, N R, 3], the rows flow g reads.  One launch forward (gather + softmax + weighted sum + layout), two
    backward (csrc/train_glue.hip); replaces GatherRowsFn + SoftmaxWsumFn + a transposing copy."""

    @staticmethod
    def forward(ctx, w, z, idx8, R, csr=None):
        lib = _lib.load()
        ctx.csr = csr                                          # deterministic mode: (off, edge) of pf_knn_csr(idx8)
        w, z, idx8 = w.contiguous(), z.contiguous(), idx8.contiguous()
        B, N, _ = z.shape
        T, K, ldw = w.shape
        a = torch.empty((T, K, R), dtype=torch.float32, device=w.device)
        u = torch.empty((B, N * R, 3), dtype=torch.float32, device=w.device)
        _lib.check(lib.pf_interp_wsum_fwd(w.data_ptr(), ldw, z.data_ptr(), idx8.data_ptr(), N, K, R, T, a.data_ptr(), u.data_ptr(),
                                          _stream()), "pf_interp_wsum_fwd")
        ctx.save_for_backward(a, z, idx8)
        ctx.dims = (N, K, R, ldw, T)
        return u

    @staticmethod
    def backward(ctx, du):
        lib = _lib.load()
        a, z, idx8 = ctx.saved_tensors
        N, K, R, ldw, T = ctx.dims
        du = du.contiguous()
        dw = torch.empty((T, K, ldw), dtype=torch.float32, device=du.device)
        dz = torch.empty_like(z)
        csr = ctx.csr if _DET else None
        if csr is not None:                                   # dz as an ordered gather over the sorted transposed lists
            _lib.check(lib.pf_interp_wsum_bwd_det(a.data_ptr(), z.data_ptr(), idx8.data_ptr(), du.data_ptr(), N, K, R, ldw, T, dw.data_ptr(),
                                                  dz.data_ptr(), csr[0].data_ptr(), csr[1].data_ptr(), _stream()), "pf_interp_wsum_bwd_det")
        else:
            _lib.check(lib.pf_interp_wsum_bwd(a.data_ptr(), z.data_ptr(), idx8.data_ptr(), du.data_ptr(), N, K, R, ldw, T, dw.data_ptr(),
                                              dz.data_ptr(), _stream()), "pf_interp_wsum_bwd")
        return dw, dz, None, None, None


def _det_inv3(W: Tensor):
    """(det, inverse) of a 3x3 matrix in closed form (cross products), differentiable.  torch.slogdet / torch.inverse go
    through a LAPACK-style solver that synchronises with the host, which a captured training step cannot do."""
    r0, r1, r2 = W[0], W[1], W[2]
    c0, c1, c2 = torch.linalg.cross(r1, r2), torch.linalg.cross(r2, r0), torch.linalg.cross(r0, r1)
    det = torch.dot(r0, c0)
    return det, torch.stack([c0, c1, c2], dim=1) / det


def _colsum3(rows: Tensor) -> Tensor:
    """[R,3] -> [3] column sums (HIP, deterministic)."""
    lib = _lib.load()
    R = rows.shape[0]
    out = torch.empty((3,), dtype=torch.float32, device=rows.device)
    ws = _ws(rows.device, 2 * lib.pf_bn_chunks(R) * 3)
    _lib.check(lib.pf_colsum(rows.data_ptr(), R, 3, out.data_ptr(), ws.data_ptr(), _stream()), "pf_colsum")
    return out


class ActNormFn(Function):
    """y = x exp(logs) + bias  (inv=0, normalize.py:34)   or   y = (x - bias) exp(-logs)  (inv=1, normalize.py:41)."""

    @staticmethod
    def forward(ctx, x, logs, bias, inv):
        lib = _lib.load()
        x = x.contiguous()
        lg, bs = logs.reshape(3).contiguous(), bias.reshape(3).contiguous()
        R = x.numel() // 3
        y = torch.empty_like(x)
        _lib.check(lib.pf_actnorm_fwd(x.data_ptr(), lg.data_ptr(), bs.data_ptr(), inv, R, y.data_ptr(), _stream()), "pf_actnorm_fwd")
        ctx.save_for_backward(x, lg, bs)
        ctx.inv, ctx.pshape = inv, logs.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, lg, bs = ctx.saved_tensors
        R = x.numel() // 3
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        gl = torch.empty((R, 3), dtype=torch.float32, device=x.device)
        gb = torch.empty((R, 3), dtype=torch.float32, device=x.device)
        _lib.check(lib.pf_actnorm_bwd(x.data_ptr(), dy.data_ptr(), lg.data_ptr(), bs.data_ptr(), ctx.inv, R, dx.data_ptr(),
                                      gl.data_ptr(), gb.data_ptr(), _stream()), "pf_actnorm_bwd")
        return dx, _colsum3(gl).view(ctx.pshape), _colsum3(gb).view(ctx.pshape), None


class CoupleInjectFn(Function):
    """h2 = y[td:] - o ; v = reverse(cat[h1,h2]) ; out = (v - t) exp(-s)   (coupling.py:55-58,114-118,132-137; permutate.py:77)."""

    @staticmethod
    def forward(ctx, y, o, s, t, td):
        lib = _lib.load()
        y, o, s, t = y.contiguous(), o.contiguous(), s.contiguous(), t.contiguous()
        R = y.numel() // 3
        out = torch.empty_like(y)
        _lib.check(lib.pf_couple_inject_fwd(y.data_ptr(), o.data_ptr(), s.data_ptr(), t.data_ptr(), td, R, out.data_ptr(), _stream()),
                   "pf_couple_inject_fwd")
        ctx.save_for_backward(out, s)
        ctx.td, ctx.oshape = td, o.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        out, s = ctx.saved_tensors
        R = out.numel() // 3
        dout = dout.contiguous()
        dy, ds, dt = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
        do = torch.empty(ctx.oshape, dtype=torch.float32, device=out.device)
        _lib.check(lib.pf_couple_inject_bwd(out.data_ptr(), dout.data_ptr(), s.data_ptr(), ctx.td, R, dy.data_ptr(), do.data_ptr(),
                                            ds.data_ptr(), dt.data_ptr(), _stream()), "pf_couple_inject_bwd")
        return dy, do, ds, dt, None


class InjectInvFn(Function):
    """v = reverse(u exp(s) + t)   (coupling.py:147-149; permutate.py:79)."""

    @staticmethod
    def forward(ctx, u, s, t):
        lib = _lib.load()
        u, s, t = u.contiguous(), s.contiguous(), t.contiguous()
        R = u.numel() // 3
        v = torch.empty_like(u)
        _lib.check(lib.pf_inject_inv_fwd(u.data_ptr(), s.data_ptr(), t.data_ptr(), R, v.data_ptr(), _stream()), "pf_inject_inv_fwd")
        ctx.save_for_backward(u, s)
        return v

    @staticmethod
    def backward(ctx, dv):
        lib = _lib.load()
        u, s = ctx.saved_tensors
        R = u.numel() // 3
        dv = dv.contiguous()
        du, ds, dt = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
        _lib.check(lib.pf_inject_inv_bwd(u.data_ptr(), s.data_ptr(), dv.data_ptr(), R, du.data_ptr(), ds.data_ptr(), dt.data_ptr(),
                                         _stream()), "pf_inject_inv_bwd")
        return du, ds, dt


class CoupleAddFn(Function):
    """out = cat[v[:td], v[td:] + o]   (coupling.py:82-85)."""

    @staticmethod
    def forward(ctx, v, o, td):
        lib = _lib.load()
        v, o = v.contiguous(), o.contiguous()
        R = v.numel() // 3
        out = torch.empty_like(v)
        _lib.check(lib.pf_couple_add(v.data_ptr(), o.data_ptr(), td, R, out.data_ptr(), _stream()), "pf_couple_add")
        ctx.td, ctx.oshape = td, o.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        g = g.contiguous()
        R = g.numel() // 3
        do = torch.empty(ctx.oshape, dtype=torch.float32, device=g.device)
        _lib.check(lib.pf_slice_tail(g.data_ptr(), ctx.td, R, do.data_ptr(), _stream()), "pf_slice_tail")
        return g, do, None


class BatchSumFn(Function):
    """x [B, ...] -> [B]: mode 0 = sum, mode 1 = sum of -0.5 (x^2 + log 2 pi)  (probs.py:73-75,87-93)."""

    @staticmethod
    def forward(ctx, x, mode):
        lib = _lib.load()
        x = x.contiguous()
        B = x.shape[0]
        M = x.numel() // B
        out = torch.empty((B,), dtype=torch.float32, device=x.device)
        _lib.check(lib.pf_batch_sum_fwd(x.data_ptr(), B, M, mode, out.data_ptr(), _stream()), "pf_batch_sum_fwd")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        B = x.shape[0]
        M = x.numel() // B
        g = g.contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.pf_batch_sum_bwd(x.data_ptr(), g.data_ptr(), B, M, ctx.mode, dx.data_ptr(), _stream()), "pf_batch_sum_bwd")
        return dx, None


# ----------------------------------------------------------------------------------------------------
# train-mode network forward (differentiable)
# ----------------------------------------------------------------------------------------------------
_UNFOLDED = os.environ.get("PF_TRAIN_FOLD", "1") == "0"


def edgeconv_train_unfolded(p, x: Tensor, idx: Tensor, pooling: bool = True) -> Tensor:
    """FeatureExtractUnit in train mode, the reference formulation on the materialised edge feature
    (interpflow.py:223-248); kept for A/B checks of the folded version below (PF_TRAIN_FOLD=0)."""
    B, N, _ = x.shape
    K = idx.shape[-1]
    f = EdgeFeatureFn.apply(x, idx)
    for seq in p.convs:
        conv, bn = seq[0], seq[1]
        y = linear(f, conv.weight, conv.bias)
        f = torch.cat([f, bn_lrelu(y, bn, 0.05)], dim=1)
    y = linear(f, p.conv_out.weight, p.conv_out.bias)
    if not pooling:
        return y
    return MaxPoolKFn.apply(y, K).view(B, N, -1)


def edgeconv_train(p, x: Tensor, idx: Tensor, pooling: bool = True, csr=None, persistent: bool = False, prefold=None,
                   tap: bool = False) -> Tensor:
    """FeatureExtractUnit in train mode (interpflow.py:234-248). x [B,N,C]; returns [B,N,odim] or [B*N*K, odim].

    Same algebra as the inference path's edge-feature fold (packing.fold_edgeconv): every conv of the dense block sees
    the edge feature [x_i; x_j; x_j - x_i] only through  (W1 - W3) x_i + (W2 + W3) x_j,  so that part of ALL five convs
    is one GEMM on the B*N points (instead of five on the B*N*K edges with 3C input channels) followed by a
    repeat / gather / add; only the growth-feature columns run per edge.  5.4x fewer MACs at C = 128 and the
    [B*N*K, 3C] edge tensor is never materialised.  Gradients reach W through the slices, x through the point GEMM
    and the gather's scatter-add - exact algebra, same results up to fp32 rounding."""
    if _UNFOLDED:
        return edgeconv_train_unfolded(p, x, idx, pooling)
    if _FUSED and _ec_fused_supported(p, x, idx, pooling):
        return edgeconv_train_fused(p, x, idx, pooling, csr, persistent, prefold=prefold, tap=tap)   # tap: -> (out, x again)
    B, N, C = x.shape
    K = idx.shape[-1]
    convs = [seq[0] for seq in p.convs] + [p.conv_out]
    Ws = [c.weight.reshape(c.weight.shape[0], -1) for c in convs]
    Wp = torch.cat([w[:, :C] - w[:, 2 * C:3 * C] for w in Ws], dim=0)            # acts on x_i
    Wq = torch.cat([w[:, C:2 * C] + w[:, 2 * C:3 * C] for w in Ws], dim=0)        # acts on x_j
    S = Wp.shape[0]
    bias = torch.cat([c.bias for c in convs] + [torch.zeros(S, dtype=torch.float32, device=x.device)])
    pq = linear(x.reshape(B * N, C), torch.cat([Wp, Wq], dim=0), bias)           # [B*N, 2S] = P (+ bias) | Q
    Pp, Qp = torch.split(pq, [S, S], dim=1)
    E = RepeatRowsFn.apply(Pp.reshape(B, N, S), K).reshape(B * N * K, S) \
        + GatherRowsFn.apply(Qp.reshape(B, N, S), idx)                            # P[i] + Q[j] per edge
    # split (not five slices): its backward is ONE concatenation instead of five zero-filled [B*N*K, S] tensors + adds
    Es = torch.split(E, [w.shape[0] for w in Ws], dim=1)
    feats: List[Tensor] = []
    for t, seq in enumerate(p.convs):
        y = Es[t]
        if feats:
            y = y + linear(feats[0] if len(feats) == 1 else torch.cat(feats, dim=1), Ws[t][:, 3 * C:])
        feats.append(bn_lrelu(y, seq[1], 0.05))
    y = Es[-1] + linear(torch.cat(feats, dim=1), Ws[-1][:, 3 * C:])
    if not pooling:
        return y.contiguous()
    return MaxPoolKFn.apply(y.contiguous(), K).view(B, N, -1)


class EdgeConvUnitFn(Function):
    """One FeatureExtractUnit in train mode as ~11 launches forward / ~20 backward (csrc/train_fused.hip: the folded edge
    feature, BatchNorm applied on load by the consumer of each layer, statistics in the GEMM epilogues, max-pool in the
    accumulator layout).  Same function and gradients as `edgeconv_train` (interpflow.py:190-248), which stays as the
    A/B reference (PF_TRAIN_FUSED=0) and as the SyncBN path."""

    @staticmethod
    def _desc(x, idx, cfg, Ws, bs, gammas, betas):
        K, g, nconv, odim, pooling, slope, eps, momentum, rmeans, rvars = cfg[:10]
        B, N, C = x.shape
        d = _lib.PfEcTrain()
        d.B, d.N, d.K, d.C, d.growth, d.nconv, d.odim, d.pooling = B, N, K, C, g, nconv, odim, int(pooling)
        d.slope, d.eps, d.momentum = slope, eps, momentum
        d.x, d.idx = x.data_ptr(), idx.data_ptr()
        for t in range(nconv + 1):
            d.W[t], d.bias[t] = Ws[t].data_ptr(), bs[t].data_ptr()
        for t in range(nconv):
            d.gamma[t], d.beta[t] = gammas[t].data_ptr(), betas[t].data_ptr()
            d.run_mean[t] = rmeans[t].data_ptr() if rmeans[t] is not None else None
            d.run_var[t] = rvars[t].data_ptr() if rvars[t] is not None else None
        return d

    @staticmethod
    def forward(ctx, x, idx, cfg, *params):
        lib = _lib.load()
        K, g, nconv, odim, pooling = cfg[:5]
        nc1 = nconv + 1
        Ws = [w.contiguous() for w in params[:nc1]]
        bs = [b.contiguous() for b in params[nc1:2 * nc1]]
        gammas = [t.contiguous() for t in params[2 * nc1:2 * nc1 + nconv]]
        betas = [t.contiguous() for t in params[2 * nc1 + nconv:]]
        x_in = x
        x = x.contiguous()
        B, N, C = x.shape
        T, E, GT = B * N, B * N * K, g * nconv
        S = GT + odim
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        pre = cfg[14] if len(cfg) > 14 else None              # (Wpq, bpq) folded by ec_prefold for this forward
        if pre is not None and tuple(pre[0].shape) != (2 * S, C):
            raise ValueError("EdgeConvUnitFn: prefolded weights of another unit")
        Wpq, bpq = pre if pre is not None else (torch.empty((2 * S, C), **f32), torch.empty((2 * S,), **f32))
        PQ, Y, aff = torch.empty((T, 2 * S), **f32), torch.empty((E, GT), **f32), torch.empty((4, GT), **f32)
        out = torch.empty((T if pooling else E, odim), **f32)
        arg = torch.empty((T, odim), dtype=torch.uint8, device=dev) if pooling else None
        d = EdgeConvUnitFn._desc(x, idx, cfg, Ws, bs, gammas, betas)
        d.Wpq, d.bpq, d.PQ, d.Y, d.aff, d.out = (Wpq.data_ptr(), bpq.data_ptr(), PQ.data_ptr(), Y.data_ptr(), aff.data_ptr(),
                                                 out.data_ptr())
        d.arg = arg.data_ptr() if pooling else None
        need = lib.pf_ec_train_ws_floats(ctypes.byref(d))
        if need < 0:
            raise _lib.PuflowHipError(f"pf_ec_train: unsupported unit shape (K={K}, growth={g}, nconv={nconv}, odim={odim})")
        ws = _ws(dev, need)
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
        d.stat = _stat(dev).data_ptr()
        if len(cfg) > 11 and cfg[11] and not _DET:            # the whole forward as one persistent launch where the library can
            d.flags, d.sync = 1, _sync_words(dev).data_ptr()
        if _DET:
            d.flags |= 2                                      # PF_TRAIN_DETERMINISTIC
        if pre is not None:
            d.flags |= 4                                      # PF_EC_PREFOLDED
        if len(cfg) > 12 and cfg[12]:                         # SyncBN: statistics over all ranks (fixed at forward time: the
            _attach_sync(d, dev)                              # backward runs after the sync_bn() scope has ended)
        _lib.check(lib.pf_ec_train_fwd(ctypes.byref(d), _stream()), "pf_ec_train_fwd")
        ctx.cfg = cfg
        ctx.has_arg = pooling
        ctx.save_for_backward(x, idx, Wpq, PQ, Y, aff, *(() if arg is None else (arg,)), *Ws, *gammas)
        res = out.view(B, N, odim) if pooling else out
        if len(cfg) > 15 and cfg[15]:
            # tap: x again, as a second output for x's OTHER consumer - that consumer's gradient then arrives HERE (dtap) and is
            # added in the epilogue of the dx GEMM (PfEcTrain.dx_add) instead of by a launch of autograd's own
            return res, x_in.view_as(x_in)
        return res

    @staticmethod
    def backward(ctx, dout, dtap=None):
        lib = _lib.load()
        cfg = ctx.cfg
        K, g, nconv, odim, pooling = cfg[:5]
        nc1 = nconv + 1
        sv = list(ctx.saved_tensors)
        x, idx, Wpq, PQ, Y, aff = sv[:6]
        arg = sv[6] if ctx.has_arg else None
        rest = sv[7 if ctx.has_arg else 6:]
        Ws, gammas = rest[:nc1], rest[nc1:]
        B, N, C = x.shape
        T, E, GT = B * N, B * N * K, g * nconv
        S = GT + odim
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dout = dout.contiguous()
        d = EdgeConvUnitFn._desc(x, idx, cfg, Ws, Ws, gammas, gammas)      # biases / betas are not read by the backward
        d.Wpq, d.PQ, d.Y, d.aff = Wpq.data_ptr(), PQ.data_ptr(), Y.data_ptr(), aff.data_ptr()
        d.arg = arg.data_ptr() if arg is not None else None
        d.dout = dout.data_ptr()
        dA, dPQ = torch.empty((E, GT), **f32), torch.empty((T, 2 * S), **f32)
        coef, dWpq = torch.empty((2, GT), **f32), torch.empty((2 * S, C), **f32)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dWs = [torch.empty_like(w) for w in Ws]
        dbs = [torch.empty((w.shape[0],), **f32) for w in Ws]
        dgs = [torch.empty((g,), **f32) for _ in range(nconv)]
        dbe = [torch.empty((g,), **f32) for _ in range(nconv)]
        d.dA, d.dPQ, d.coef, d.dWpq = dA.data_ptr(), dPQ.data_ptr(), coef.data_ptr(), dWpq.data_ptr()
        d.dx = dx.data_ptr() if dx is not None else None
        if dtap is not None and dx is not None:
            dtap = dtap.contiguous()
            if dtap.shape != x.shape or dtap.dtype != torch.float32:
                raise ValueError("EdgeConvUnitFn: gradient of the tap has another shape than x")
            d.dx_add = dtap.data_ptr()
        for t in range(nc1):
            d.dW[t], d.dbias[t] = dWs[t].data_ptr(), dbs[t].data_ptr()
        for t in range(nconv):
            d.dgamma[t], d.dbeta[t] = dgs[t].data_ptr(), dbe[t].data_ptr()
        need = lib.pf_ec_train_ws_floats(ctypes.byref(d))
        ws = _ws(dev, need)
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
        d.stat = _stat(dev).data_ptr()
        csr = cfg[10] if len(cfg) > 10 else None
        if csr is not None:                                   # transposed neighbour lists: dQ as a gather, no float atomics
            d.csr_off, d.csr_edge = csr[0].data_ptr(), csr[1].data_ptr()
        if len(cfg) > 11 and cfg[11] and not _DET:            # the dense block's backward as one persistent launch (see forward)
            d.flags, d.sync = 1, _sync_words(dev).data_ptr()
        if _DET:
            d.flags |= 2
        if len(cfg) > 12 and cfg[12]:
            _attach_sync(d, dev)
        dwst = None if (len(cfg) > 13 and cfg[13]) else _dw_begin(dev, sv, dout, dA, dPQ, coef, dWpq, ws)
        if dwst is not None:                                  # weight gradients on their own stream, with their own workspace
            ws2 = _dw_ws(dwst, dev, need)
            d.ws_dw, d.ws_dw_floats = ws2.data_ptr(), ws2.numel()
        with _dw_call(dwst):
            _lib.check(lib.pf_ec_train_bwd(ctypes.byref(d), _stream()), "pf_ec_train_bwd")
        return (dx, None, None, *dWs, *dbs, *dgs, *dbe)


_COUNTER = {}


def _counter(dev) -> Tensor:
    """One zero-initialised 32-bit word per (device, stream): the arrival counter of the in-kernel grid reductions (reset by
    the workgroup that uses it last)."""
    key = (dev, _stream())
    t = _COUNTER.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        _COUNTER[key] = t
    return t


def _ptr(t):
    return t.data_ptr() if t is not None else None


class FlowParamsFn(Function):
    """W [3,3], logs [...,3] -> (W^-1 [3,3], ld [1] = (sum(logs) + log|det W|) n): the parameter-only scalars of a flow block
    (normalize.py:34-36, permutate.py:118-124) in one one-thread kernel instead of ~25 tiny torch launches."""

    @staticmethod
    def forward(ctx, W, logs, n):
        lib = _lib.load()
        W, lg = W.contiguous(), logs.reshape(3).contiguous()
        Winv = torch.empty_like(W)
        ld = torch.empty((1,), dtype=torch.float32, device=W.device)
        _lib.check(lib.pf_flow_params_fwd(W.data_ptr(), lg.data_ptr(), float(n), Winv.data_ptr(), ld.data_ptr(), _stream()),
                   "pf_flow_params_fwd")
        ctx.save_for_backward(Winv)
        ctx.n, ctx.lshape = float(n), logs.shape
        return Winv, ld

    @staticmethod
    def backward(ctx, dWinv, dld):
        lib = _lib.load()
        (Winv,) = ctx.saved_tensors
        dW = torch.empty_like(Winv)
        dlogs = torch.empty((3,), dtype=torch.float32, device=Winv.device)
        dWinv = dWinv.contiguous() if dWinv is not None else None
        dld = dld.contiguous() if dld is not None else None
        _lib.check(lib.pf_flow_params_bwd(Winv.data_ptr(), _ptr(dWinv), _ptr(dld), ctx.n, dW.data_ptr(), dlogs.data_ptr(),
                                          _stream()), "pf_flow_params_bwd")
        return dW, dlogs.view(ctx.lshape), None


class FlowAffineFn(Function):
    """inv=0: y = M (x e^logs + bias)  (ActNorm, then the 3x3 linear);  inv=1: y = (M [x_head, x_tail + o] - bias) e^-logs
    (coupling shift, inverse linear, inverse ActNorm).  One launch each way; the 15 parameter-gradient sums are reduced
    inside the backward kernel (csrc/train_flow.hip)."""

    @staticmethod
    def forward(ctx, x, o, td, logs, bias, M, inv):
        lib = _lib.load()
        x = x.contiguous()
        o = o.contiguous() if o is not None else None
        lg, bs, M = logs.reshape(3).contiguous(), bias.reshape(3).contiguous(), M.contiguous()
        R = x.numel() // 3
        y = torch.empty_like(x)
        _lib.check(lib.pf_flow_affine_fwd(x.data_ptr(), _ptr(o), td, lg.data_ptr(), bs.data_ptr(), M.data_ptr(), inv, R,
                                          y.data_ptr(), _stream()), "pf_flow_affine_fwd")
        ctx.save_for_backward(x, lg, bs, M, *(() if o is None else (o,)))
        ctx.cfg = (td, inv, logs.shape, o is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        td, inv, pshape, has_o = ctx.cfg
        sv = ctx.saved_tensors
        x, lg, bs, M = sv[:4]
        o = sv[4] if has_o else None
        R = x.numel() // 3
        dy = dy.contiguous()
        dev = x.device
        dx = torch.empty_like(x)
        do = torch.empty_like(o) if has_o else None
        dl, db = torch.empty((3,), dtype=torch.float32, device=dev), torch.empty((3,), dtype=torch.float32, device=dev)
        dM = torch.empty_like(M)
        ws = _ws(dev, 256 * 16)
        _lib.check(lib.pf_flow_affine_bwd(x.data_ptr(), _ptr(o), td, lg.data_ptr(), bs.data_ptr(), M.data_ptr(), inv, R,
                                          dy.data_ptr(), dx.data_ptr(), _ptr(do), dl.data_ptr(), db.data_ptr(), dM.data_ptr(),
                                          ws.data_ptr(), _counter(dev).data_ptr(), _stream()), "pf_flow_affine_bwd")
        return dx, do, None, dl.view(pshape), db.view(pshape), dM, None


class CoupleInject2Fn(Function):
    """CoupleInjectFn that also returns sum(s) (the injector's log-det term, coupling.py:137) from the same launch."""

    @staticmethod
    def forward(ctx, y, o, s, t, td):
        lib = _lib.load()
        y, o, s, t = y.contiguous(), o.contiguous(), s.contiguous(), t.contiguous()
        R = y.numel() // 3
        dev = y.device
        out = torch.empty_like(y)
        ssum = torch.empty((1,), dtype=torch.float32, device=dev)
        ws = _ws(dev, 256 * 16)
        _lib.check(lib.pf_couple_inject2_fwd(y.data_ptr(), o.data_ptr(), s.data_ptr(), t.data_ptr(), td, R, out.data_ptr(),
                                             ssum.data_ptr(), ws.data_ptr(), _counter(dev).data_ptr(), _stream()),
                   "pf_couple_inject2_fwd")
        ctx.save_for_backward(out, s)
        ctx.td, ctx.oshape = td, o.shape
        return out, ssum

    @staticmethod
    def backward(ctx, dout, dssum):
        lib = _lib.load()
        out, s = ctx.saved_tensors
        R = out.numel() // 3
        dout = dout.contiguous()
        dssum = dssum.contiguous() if dssum is not None else None
        dy, ds, dt = torch.empty_like(out), torch.empty_like(s), torch.empty_like(s)
        do = torch.empty(ctx.oshape, dtype=torch.float32, device=out.device)
        _lib.check(lib.pf_couple_inject2_bwd(out.data_ptr(), dout.data_ptr(), _ptr(dssum), s.data_ptr(), ctx.td, R, dy.data_ptr(),
                                             do.data_ptr(), ds.data_ptr(), dt.data_ptr(), _stream()), "pf_couple_inject2_bwd")
        return dy, do, ds, dt, None


class InjectInv2Fn(Function):
    """v = reverse(u e^s + t) with s, t [T,3] of the ORIGINAL points and u [T*R,3]: the repeat_interleave of the reference
    (interpflow.py:319) happens in the index, its backward (sum over the R rows) in the same kernel."""

    @staticmethod
    def forward(ctx, u, s, t, Rr):
        lib = _lib.load()
        u, s, t = u.contiguous(), s.contiguous(), t.contiguous()
        R = u.numel() // 3
        v = torch.empty_like(u)
        _lib.check(lib.pf_inject_inv2_fwd(u.data_ptr(), s.data_ptr(), t.data_ptr(), Rr, R, v.data_ptr(), _stream()),
                   "pf_inject_inv2_fwd")
        ctx.save_for_backward(u, s)
        ctx.Rr = Rr
        return v

    @staticmethod
    def backward(ctx, dv):
        lib = _lib.load()
        u, s = ctx.saved_tensors
        R = u.numel() // 3
        dv = dv.contiguous()
        du, ds, dt = torch.empty_like(u), torch.empty_like(s), torch.empty_like(s)
        _lib.check(lib.pf_inject_inv2_bwd(u.data_ptr(), s.data_ptr(), dv.data_ptr(), ctx.Rr, R, du.data_ptr(), ds.data_ptr(),
                                          dt.data_ptr(), _stream()), "pf_inject_inv2_bwd")
        return du, ds, dt, None


_FC_IMG = {}                     # device -> (key, image) of the last flow-chain forward (FlowChainFn.forward)
_FC_SCOPE = [0, 0]               # [id of the forward_train call in progress (0: none), calls so far]
_IMG_SHARE = os.environ.get("PF_TRAIN_IMG_SHARE", "1") != "0"


class FlowChainFn(Function):
    """All flow blocks of one direction as ONE autograd node: two launches forward, four backward (csrc/train_flowchain.hip).
    apply(inv, R, n_ld, ccs, x, cflat, st, Bsz, *[logs, bias, W, w0, w2, b2, w4, b4] per block)
      ccs    conditioning channels per block; cflat = the blocks' conditioning features [T, cc_i], flattened and concatenated
             (ONE tensor: its three consumers cost two gradient additions instead of twelve)
      st     [2 nb, T, 3]: injector scale (2 i) and shift (2 i + 1) of block i per ORIGINAL point
      inv = 0 (PointInterpFlow.f, interpflow.py:302-310): x [B,N,3] -> (z, ssum [nb] = sum(s_i), ld [nb] = (sum(logs_i) + log|det W_i|) n_ld);
              with Bsz > 0 instead (z, logp [1]) - the log-likelihood -mean_b(log N(z_b) + sum_i (ld_i - sum(s_i)[b])) of
              interpflow.py:327-337 from the kernel's own epilogue, its backward folded into the chain kernel
      inv = 1 (PointInterpFlow.g, interpflow.py:312-321): u [B,N R,3] -> x, blocks in reverse order
    Replaces, per block, FlowParamsFn + FlowAffineFn + MlpFn + CoupleInject2Fn / InjectInv2Fn and the gradient-accumulation adds
    autograd inserted between them."""

    @staticmethod
    def _desc(inv, R, n_ld, ccs, x, cflat, st, prm):
        nb = len(ccs)
        d = _lib.PfFlowChain()
        d.nb, d.rows, d.R, d.inv, d.n_ld = nb, x.numel() // 3, R, inv, float(n_ld)
        d.x = x.data_ptr()
        T = d.rows // R
        off = 0
        for i in range(nb):
            lg, bi, W, w0, w2, b2, w4, b4 = prm[8 * i:8 * i + 8]
            d.cc[i] = ccs[i]
            d.td[i] = w0.shape[1] - ccs[i]
            d.c[i] = cflat.data_ptr() + 4 * off
            off += T * ccs[i]
            d.s[i], d.t[i] = st[2 * i].data_ptr(), st[2 * i + 1].data_ptr()
            d.logs[i], d.bias[i], d.W[i] = lg.data_ptr(), bi.data_ptr(), W.data_ptr()
            d.w0[i], d.w2[i], d.b2[i], d.w4[i], d.b4[i] = w0.data_ptr(), w2.data_ptr(), b2.data_ptr(), w4.data_ptr(), b4.data_ptr()
        if cflat.numel() != off or tuple(st.shape) != (2 * nb, T, 3):
            raise ValueError("FlowChainFn: conditioning tensors do not match the row count")
        return d

    @staticmethod
    def forward(ctx, inv, R, n_ld, ccs, x, cflat, st, Bsz, *prm):
        lib = _lib.load()
        x, cflat, st = x.contiguous(), cflat.contiguous(), st.contiguous()
        prm = [t.contiguous() for t in prm]
        nb = len(ccs)
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        d = FlowChainFn._desc(inv, R, n_ld, ccs, x, cflat, st, prm)
        rows = d.rows
        keep = torch.empty((2, nb, rows, 3), **f32)                      # block inputs | y (f) / v (g)
        hh = torch.empty((2, nb, rows, 64), **f32)
        o = torch.empty((nb, rows, 2), **f32) if inv else None
        out = torch.empty_like(x)
        ssum, ld = torch.empty((nb,), **f32), torch.empty((nb,), **f32)   # sum(s) | log-det term, per block
        d.pin, d.mid, d.h1, d.h2, d.out = keep[0].data_ptr(), keep[1].data_ptr(), hh[0].data_ptr(), hh[1].data_ptr(), out.data_ptr()
        d.o = _ptr(o)
        d.ssum, d.ld = ssum.data_ptr(), ld.data_ptr()
        logp = None
        if Bsz and not inv:
            logp = torch.empty((1,), **f32)
            d.logp, d.Bsz = logp.data_ptr(), int(Bsz)
        d.part = _ws(dev, (nb + 1) * ((rows + 15) // 16)).data_ptr()
        d.counter = _counter(dev).data_ptr()
        # packed weights (the conditioner nets' LDS images), kept for the backward.  They do not depend on the direction: the g chain
        # of a forward takes the image the f chain packed from the same parameters on the same stream (one pack launch fewer)
        key = (_FC_SCOPE[0], tuple(int(c) for c in ccs), _stream(), tuple((t.data_ptr(), t._version) for t in prm))
        cached = _FC_IMG.get(dev) if (_IMG_SHARE and _FC_SCOPE[0]) else None      # only inside one forward_train call
        if cached is not None and cached[0] == key:
            img = cached[1]
            d.img_ready = 1
        else:
            img = torch.empty((lib.pf_flowchain_img_floats(ctypes.byref(d)),), **f32)
            _FC_IMG[dev] = (key, img)
        d.img = img.data_ptr()
        _lib.check(lib.pf_flowchain_fwd(ctypes.byref(d), _stream()), "pf_flowchain_fwd")
        ctx.cfg = (inv, R, float(n_ld), tuple(ccs), [t.shape for t in prm], int(Bsz) if logp is not None else 0)
        ctx.save_for_backward(x, out, keep, hh, img, cflat, st, *(() if o is None else (o,)), *prm)
        if inv:
            return out
        if logp is not None:
            return out, logp
        return out, ssum, ld

    @staticmethod
    def backward(ctx, dout, dssum=None, dld=None):
        lib = _lib.load()
        inv, R, n_ld, ccs, pshapes, Bsz = ctx.cfg
        dlogp = None
        if Bsz:                                            # outputs were (z, logp)
            dlogp, dssum = (dssum.contiguous().view(1) if dssum is not None else None), None
        nb = len(ccs)
        sv = list(ctx.saved_tensors)
        x, out, keep, hh, img, cflat, st = sv[:7]
        o, prm = (sv[7], sv[8:]) if inv else (None, sv[7:])
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        d = FlowChainFn._desc(inv, R, n_ld, ccs, x, cflat, st, prm)
        rows = d.rows
        T = rows // R
        d.pin, d.mid, d.h1, d.h2, d.out = keep[0].data_ptr(), keep[1].data_ptr(), hh[0].data_ptr(), hh[1].data_ptr(), out.data_ptr()
        d.o = _ptr(o)
        d.img = img.data_ptr()
        dout = dout.contiguous() if dout is not None else None
        dssum = dssum.contiguous() if dssum is not None else None
        dld = dld.contiguous() if dld is not None else None
        if dout is None and dlogp is None:
            dout = torch.zeros_like(x)
        d.dout, d.dssum, d.dld, d.dlogp, d.Bsz = _ptr(dout), _ptr(dssum), _ptr(dld), _ptr(dlogp), Bsz
        dx = torch.empty_like(x) if ctx.needs_input_grad[4] else None
        d.dx = _ptr(dx)
        dcflat = torch.empty_like(cflat)
        dst = torch.empty_like(st)
        dz = torch.empty((2, nb, rows, 64), **f32)
        dob = torch.empty((nb, rows, 2), **f32)
        d.dz1, d.dz2, d.dob = dz[0].data_ptr(), dz[1].data_ptr(), dob.data_ptr()
        dzs = None
        if R > 1 and _DZSUM:
            # the first hidden layer's gradient summed over the R rows that share a conditioning row: the weight-gradient launch
            # then runs that layer's conditioning columns (cc of cc + td) over rows / R summed rows (PF_MLP_DW_DZSUM)
            dzs = torch.empty((nb, T, 64), **f32)
            d.dz1s = dzs.data_ptr()
        sizes = [int(p.numel()) for p in prm]
        flat = torch.empty((sum(sizes),), **f32)                         # every parameter gradient of the chain, one buffer
        gp, off = [], 0
        for n_ in sizes:
            gp.append(flat[off:off + n_])
            off += n_
        coff = 0
        for i in range(nb):
            d.dc[i] = dcflat.data_ptr() + 4 * coff
            coff += T * ccs[i]
            d.ds[i], d.dt[i] = dst[2 * i].data_ptr(), dst[2 * i + 1].data_ptr()
            g = gp[8 * i:8 * i + 8]
            d.dlogs[i], d.dbias[i], d.dW[i] = g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr()
            d.dw0[i], d.dw2[i], d.db2[i], d.dw4[i], d.db4[i] = (g[3].data_ptr(), g[4].data_ptr(), g[5].data_ptr(), g[6].data_ptr(),
                                                                 g[7].data_ptr())
        npart = lib.pf_flowchain_part_floats(ctypes.byref(d))
        need = lib.pf_flowchain_ws_floats(ctypes.byref(d))
        if need < 0 or npart < 0:
            raise _lib.PuflowHipError("pf_flowchain: unsupported shape")
        ws = _ws(dev, npart + need)
        d.part = ws.data_ptr()
        d.ws, d.ws_floats = ws.data_ptr() + 4 * npart, need
        d.dev_descs = _desc_buf(dev).data_ptr()
        # (not on the weight-gradient stream: the f and the g chain share their parameters, so autograd ADDS the two chains'
        # gradients on this stream as soon as the second one returns - before any join)
        _lib.check(lib.pf_flowchain_bwd(ctypes.byref(d), _stream()), "pf_flowchain_bwd")
        grads = [g.view(shp) for g, shp in zip(gp, pshapes)]
        return (None, None, None, None, dx, dcflat, dst, None, *grads)


_DZSUM = os.environ.get("PF_TRAIN_DZSUM", "1") != "0"         # g chain: conditioner weight gradients from replica-summed dz (FlowChainFn.backward)
_FANOUT = os.environ.get("PF_TRAIN_FANOUT", "1") != "0"       # gradients of the flattened conditioning features summed in one launch


class FanoutFn(Function):
    """apply(n, x) -> n aliases of x, one per consumer; backward: the n gradients summed in ONE launch (pf_sum_n,
    ((g0 + g1) + g2) + ...) instead of autograd's n - 1 pairwise adds over the running sum."""

    @staticmethod
    def forward(ctx, n, x):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return None, gs[0]
        g0 = gs[0]
        if (g0.numel() % 4 or len(gs) > 8 or not g0.is_cuda
                or any(g.dtype != torch.float32 or g.shape != g0.shape or g.data_ptr() % 16 for g in gs)):
            out = gs[0] + gs[1]
            for g in gs[2:]:
                out = out + g
            return None, out
        out = torch.empty_like(g0)
        ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        _lib.check(_lib.load().pf_sum_n(ptrs, len(gs), out.data_ptr(), g0.numel(), _stream()), "pf_sum_n")
        return None, out


class CondNetStackFn(Function):
    """The injector scale / shift conditioners (LinearA1D, first layer without bias, interpflow.py:22-43) of ALL flow blocks on the
    flattened conditioning features: one launch forward, four backward (csrc/train_mlp.hip, batched entry points), ONE output
    tensor.  apply(ccs, T, cflat, cflat2, *[W0, W1, b1, W2, b2 per net]) -> st [n, T, 3]; net k reads block k // 2's features.
    cflat2: None, or a second alias of cflat (FanoutFn) - the gradient through the scale nets then goes to cflat and the one
    through the shift nets to cflat2, un-added (the fan-out sums them with the other consumers' in its one launch)."""

    @staticmethod
    def _descs(ccs, T, cflat, prm, n):
        descs = (_lib.PfMlpTrain * n)()
        offs, off = [], 0
        for cc in ccs:
            offs.append(off)
            off += T * cc
        for k in range(n):
            W0, W1, b1, W2, b2 = prm[5 * k:5 * k + 5]
            d = _lib.PfMlpTrain()
            d.rows, d.nl, d.td, d.cc, d.cdiv, d.ldy = T, 3, 0, ccs[k // 2], 1, 0
            for l, w in enumerate((W0, W1, W2)):
                d.width[l] = w.shape[0]
                d.W[l] = w.data_ptr()
            d.slope[0] = d.slope[1] = 0.01
            d.chunk = 256                                   # n networks x 3 layers in one launch: long split-K chunks
            d.c = cflat.data_ptr() + 4 * offs[k // 2]
            descs[k] = d
        return descs, offs

    @staticmethod
    def forward(ctx, ccs, T, cflat, cflat2, *prm):
        lib = _lib.load()
        n = len(prm) // 5
        ctx.two = cflat2 is not None
        cflat = cflat.contiguous()
        prm = [w.contiguous() for w in prm]
        dev = cflat.device
        f32 = dict(dtype=torch.float32, device=dev)
        descs, _ = CondNetStackFn._descs(ccs, T, cflat, prm, n)
        hs = torch.empty((n, 2, T, 64), **f32)
        st = torch.empty((n, T, 3), **f32)
        for k in range(n):
            W0, W1, b1, W2, b2 = prm[5 * k:5 * k + 5]
            if W0.shape[0] != 64 or W1.shape[0] != 64 or W2.shape[0] != 3:
                raise ValueError("CondNetStackFn: unexpected conditioner shape")
            descs[k].b[1], descs[k].b[2] = b1.data_ptr(), b2.data_ptr()
            descs[k].h[0], descs[k].h[1], descs[k].out = hs[k, 0].data_ptr(), hs[k, 1].data_ptr(), st[k].data_ptr()
        _lib.check(lib.pf_mlp_train_fwd_batch(descs, n, _desc_buf(dev).data_ptr(), _stream()), "pf_mlp_train_fwd_batch")
        ctx.cfg = (tuple(ccs), T, n)
        ctx.save_for_backward(cflat, hs, *prm)
        return st

    @staticmethod
    def backward(ctx, dst):
        lib = _lib.load()
        ccs, T, n = ctx.cfg
        sv = list(ctx.saved_tensors)
        cflat, hs, prm = sv[0], sv[1], sv[2:]
        dev = cflat.device
        f32 = dict(dtype=torch.float32, device=dev)
        dst = dst.contiguous()
        descs, offs = CondNetStackFn._descs(ccs, T, cflat, prm, n)
        dz = torch.empty_like(hs)
        dc2 = torch.empty((2, cflat.numel()), **f32)                    # gradients through the scale nets | through the shift nets
        sizes = [int(p.numel()) for p in prm]
        flat = torch.empty((sum(sizes),), **f32)
        gp, off = [], 0
        for n_ in sizes:
            gp.append(flat[off:off + n_])
            off += n_
        need = [lib.pf_mlp_train_ws_floats(ctypes.byref(descs[k])) for k in range(n)]
        dwst = _dw_begin(dev, sv, dst, dz, dc2, flat)
        ws = _ws(dev, sum(need)) if dwst is None else _dw_ws(dwst, dev, sum(need))     # weight-gradient scratch only
        ddesc = _desc_buf(dev)
        if dwst is not None:                                  # the side kernels read the descriptors: this call's own copy
            ddesc = torch.empty(16 * ctypes.sizeof(_lib.PfMlpTrain), dtype=torch.uint8, device=dev)
            _DW_PASS["keep"].append((ddesc,))
        woff = 0
        for k in range(n):
            d = descs[k]
            d.h[0], d.h[1], d.dz[0], d.dz[1] = hs[k, 0].data_ptr(), hs[k, 1].data_ptr(), dz[k, 0].data_ptr(), dz[k, 1].data_ptr()
            d.dout = dst[k].data_ptr()
            d.dc = dc2[k % 2].data_ptr() + 4 * offs[k // 2]
            g = gp[5 * k:5 * k + 5]
            d.dW[0], d.dW[1], d.dW[2] = g[0].data_ptr(), g[1].data_ptr(), g[3].data_ptr()
            d.db[1], d.db[2] = g[2].data_ptr(), g[4].data_ptr()
            d.ws, d.ws_floats = ws.data_ptr() + 4 * woff, need[k]
            woff += need[k]
            descs[k] = d
        with _dw_call(dwst):
            _lib.check(lib.pf_mlp_train_bwd_batch(descs, n, ddesc.data_ptr(), _stream()), "pf_mlp_train_bwd_batch")
        if ctx.two:
            return (None, None, dc2[0], dc2[1], *[g.view(p.shape) for g, p in zip(gp, prm)])
        return (None, None, dc2[0] + dc2[1], None, *[g.view(p.shape) for g, p in zip(gp, prm)])


class MlpFn(Function):
    """2- or 3-layer point-wise MLP (LinearA1D / FeatMergeUnit, interpflow.py:22-43, 251-258) on cat[y[:, :td], c[row // cdiv]]:
    one launch forward, three backward (csrc/train_mlp.hip).  wb = W0, b0, W1, b1[, W2, b2] (None for a missing bias)."""

    @staticmethod
    def _desc(y, c, td, cdiv, slopes, Ws, bs):
        d = _lib.PfMlpTrain()
        nl = len(Ws)
        cc = c.shape[-1]
        rows = c.numel() // cc * cdiv
        d.rows, d.nl, d.td, d.cc, d.cdiv = rows, nl, td, cc, cdiv
        d.ldy = y.shape[-1] if y is not None else 0
        for l in range(nl):
            d.width[l] = Ws[l].shape[0]
            d.W[l] = Ws[l].data_ptr()
            d.b[l] = bs[l].data_ptr() if bs[l] is not None else None
        for l in range(nl - 1):
            d.slope[l] = slopes[l]
        d.y = y.data_ptr() if y is not None else None
        d.c = c.data_ptr()
        return d, rows

    @staticmethod
    def forward(ctx, y, c, td, cdiv, slopes, *wb):
        lib = _lib.load()
        Ws = [w.contiguous() for w in wb[0::2]]
        bs = [b.contiguous() if b is not None else None for b in wb[1::2]]
        y = y.contiguous() if (y is not None and td > 0) else None
        c = c.contiguous()
        d, rows = MlpFn._desc(y, c, td, cdiv, slopes, Ws, bs)
        f32 = dict(dtype=torch.float32, device=c.device)
        hs = [torch.empty((rows, Ws[l].shape[0]), **f32) for l in range(len(Ws) - 1)]
        out = torch.empty((rows, Ws[-1].shape[0]), **f32)
        for l, h in enumerate(hs):
            d.h[l] = h.data_ptr()
        d.out = out.data_ptr()
        _lib.check(lib.pf_mlp_train_fwd(ctypes.byref(d), _stream()), "pf_mlp_train_fwd")
        ctx.cfg = (td, cdiv, slopes, len(Ws), [b is not None for b in bs], y is not None)
        ctx.save_for_backward(c, *hs, *Ws, *(() if y is None else (y,)))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        td, cdiv, slopes, nl, has_b, has_y = ctx.cfg
        sv = list(ctx.saved_tensors)
        c, hs, Ws = sv[0], sv[1:nl], sv[nl:2 * nl]
        y = sv[2 * nl] if has_y else None
        dout = dout.contiguous()
        d, rows = MlpFn._desc(y, c, td, cdiv, slopes, Ws, [None] * nl)
        f32 = dict(dtype=torch.float32, device=c.device)
        dzs = [torch.empty_like(h) for h in hs]
        dy = torch.empty_like(y) if (has_y and ctx.needs_input_grad[0]) else None
        dc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        dWs = [torch.empty_like(w) for w in Ws]
        dbs = [torch.empty((w.shape[0],), **f32) if has_b[l] else None for l, w in enumerate(Ws)]
        for l in range(nl - 1):
            d.h[l], d.dz[l] = hs[l].data_ptr(), dzs[l].data_ptr()
        d.dout = dout.data_ptr()
        d.dy = dy.data_ptr() if dy is not None else None
        d.dc = dc.data_ptr() if dc is not None else None
        for l in range(nl):
            d.dW[l] = dWs[l].data_ptr()
            d.db[l] = dbs[l].data_ptr() if dbs[l] is not None else None
        need = lib.pf_mlp_train_ws_floats(ctypes.byref(d))
        if need < 0:
            raise _lib.PuflowHipError("pf_mlp_train: unsupported shape")
        dwst = _dw_begin(c.device, list(ctx.saved_tensors), dout, dzs, hs)
        ws = _ws(c.device, need) if dwst is None else _dw_ws(dwst, c.device, need)     # weight-gradient scratch only
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
        with _dw_call(dwst):
            _lib.check(lib.pf_mlp_train_bwd(ctypes.byref(d), _stream()), "pf_mlp_train_bwd")
        grads = []
        for l in range(nl):
            grads += [dWs[l], dbs[l]]
        return (dy, dc, None, None, None, *grads)


class FoldWuFn(Function):
    """(W0 [o, 2 o, 1, 1], b0, W6 [o, k6, 1, 1], b6, Wout [o, ko, 1, 1], bout) -> (W0a W6, W0a b6 + b0, W0b Wout, W0b bout) in the
    shapes of W6 / b6 / Wout / bout: WeightEstimationUnit's first conv folded into the last linear layers of its two producers
    (csrc/train_glue.hip pf_fold_wu_fwd / _bwd: fixed summation order, one launch per direction)."""

    @staticmethod
    def forward(ctx, W0, b0, W6, b6, Wout, bout):
        lib = _lib.load()
        W0, b0, W6, b6, Wout, bout = (t.contiguous() for t in (W0, b0, W6, b6, Wout, bout))
        o = W0.shape[0]
        k6, ko = W6.numel() // o, Wout.numel() // o
        assert W0.numel() == 2 * o * o and W6.shape[0] == o and Wout.shape[0] == o
        W6f, b6f, Wof, bof = torch.empty_like(W6), torch.empty_like(b6), torch.empty_like(Wout), torch.empty_like(bout)
        _lib.check(lib.pf_fold_wu_fwd(W0.data_ptr(), b0.data_ptr(), W6.data_ptr(), b6.data_ptr(), Wout.data_ptr(), bout.data_ptr(),
                                      o, k6, ko, W6f.data_ptr(), b6f.data_ptr(), Wof.data_ptr(), bof.data_ptr(), _stream()),
                   "pf_fold_wu_fwd")
        ctx.save_for_backward(W0, W6, b6, Wout, bout)
        return W6f, b6f, Wof, bof

    @staticmethod
    def backward(ctx, dW6f, db6f, dWof, dbof):
        lib = _lib.load()
        W0, W6, b6, Wout, bout = ctx.saved_tensors
        o = W0.shape[0]
        k6, ko = W6.numel() // o, Wout.numel() // o
        z = lambda g, like: torch.zeros_like(like) if g is None else g.contiguous()
        dW6f, db6f, dWof, dbof = z(dW6f, W6), z(db6f, b6), z(dWof, Wout), z(dbof, bout)
        dW0, db0 = torch.empty_like(W0), torch.empty_like(b6)
        dW6, db6, dWout, dbout = torch.empty_like(W6), torch.empty_like(b6), torch.empty_like(Wout), torch.empty_like(bout)
        _lib.check(lib.pf_fold_wu_bwd(W0.data_ptr(), W6.data_ptr(), b6.data_ptr(), Wout.data_ptr(), bout.data_ptr(), o, k6, ko,
                                      dW6f.data_ptr(), db6f.data_ptr(), dWof.data_ptr(), dbof.data_ptr(), dW0.data_ptr(),
                                      db0.data_ptr(), dW6.data_ptr(), db6.data_ptr(), dWout.data_ptr(), dbout.data_ptr(),
                                      _stream()), "pf_fold_wu_bwd")
        return dW0, db0, dW6, db6, dWout, dbout


class BnMlpFn(Function):
    """[Conv2d 1x1 + BatchNorm2d(train) + LeakyReLU] x 2 + Conv2d 1x1 on rows (DistanceEncoder / WeightEstimationUnit,
    interpflow.py:85-151) on cat[xa, xb] without building it: 3-4 launches forward, ~10 backward (csrc/train_fused.hip).
    apply(xa, xb | None, cfg, W0, b0, W1, b1, W2, b2, gamma0, beta0, gamma1, beta1)"""

    @staticmethod
    def _desc(xa, xb, cfg, Ws):
        slope, eps, momentum, rmeans, rvars = cfg[:5]
        d = _lib.PfBnMlpTrain()
        d.rows, d.nl = xa.shape[0], len(Ws)
        d.kin0a, d.kin0b = xa.shape[1], (xb.shape[1] if xb is not None else 0)
        for l, w in enumerate(Ws):
            d.width[l] = w.shape[0]
            d.W[l] = w.data_ptr()
        d.slope, d.eps, d.momentum = slope, eps, momentum
        d.xa, d.xb = xa.data_ptr(), _ptr(xb)
        return d

    @staticmethod
    def forward(ctx, xa, xb, cfg, *prm):
        lib = _lib.load()
        slope, eps, momentum, rmeans, rvars = cfg[:5]
        xa = xa.contiguous()
        xb = xb.contiguous() if xb is not None else None
        Ws = [w.contiguous() for w in prm[0:6:2]]
        bs = [b.contiguous() for b in prm[1:6:2]]
        gb = [g.contiguous() for g in prm[6:10]]
        dev = xa.device
        f32 = dict(dtype=torch.float32, device=dev)
        d = BnMlpFn._desc(xa, xb, cfg, Ws)
        ys = [torch.empty((xa.shape[0], w.shape[0]), **f32) for w in Ws]
        affs = [torch.empty((4, Ws[l].shape[0]), **f32) for l in range(2)]
        for l in range(3):
            d.b[l], d.y[l] = bs[l].data_ptr(), ys[l].data_ptr()
        for l in range(2):
            d.gamma[l], d.beta[l], d.aff[l] = gb[2 * l].data_ptr(), gb[2 * l + 1].data_ptr(), affs[l].data_ptr()
            d.run_mean[l], d.run_var[l] = _ptr(rmeans[l]), _ptr(rvars[l])
        d.stat = _stat(dev).data_ptr()
        d.flags = (2 if _DET else 0) | (4 if (len(cfg) > 6 and cfg[6]) else 0)      # 4 = PF_BNMLP_SUM_INPUTS: y[0] = xa + xb
        if len(cfg) > 5 and cfg[5]:
            _attach_sync(d, dev)
        _lib.check(lib.pf_bnmlp_train_fwd(ctypes.byref(d), _stream()), "pf_bnmlp_train_fwd")
        ctx.cfg, ctx.has_b = cfg, xb is not None
        ctx.save_for_backward(xa, *(() if xb is None else (xb,)), *Ws, *ys, *affs, gb[0], gb[2])
        return ys[2]

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        sv = list(ctx.saved_tensors)
        xa = sv[0]
        xb = sv[1] if ctx.has_b else None
        o = 2 if ctx.has_b else 1
        Ws, ys, affs, gam = sv[o:o + 3], sv[o + 3:o + 6], sv[o + 6:o + 8], sv[o + 8:o + 10]
        dev = xa.device
        f32 = dict(dtype=torch.float32, device=dev)
        dout = dout.contiguous()
        d = BnMlpFn._desc(xa, xb, ctx.cfg, Ws)
        ds = [torch.empty_like(ys[0]), torch.empty_like(ys[1])]
        coefs = [torch.empty((2, Ws[l].shape[0]), **f32) for l in range(2)]
        dxa = torch.empty_like(xa) if ctx.needs_input_grad[0] else None
        dxb = torch.empty_like(xb) if (xb is not None and ctx.needs_input_grad[1]) else None
        dWs = [torch.empty_like(w) for w in Ws]
        dbs = [torch.empty((w.shape[0],), **f32) for w in Ws]
        dgs = [torch.empty((Ws[l].shape[0],), **f32) for l in range(2)]
        dbe = [torch.empty((Ws[l].shape[0],), **f32) for l in range(2)]
        for l in range(3):
            d.y[l], d.dW[l], d.db[l] = ys[l].data_ptr(), dWs[l].data_ptr(), dbs[l].data_ptr()
        for l in range(2):
            d.gamma[l], d.beta[l] = gam[l].data_ptr(), gam[l].data_ptr()
            d.aff[l], d.d[l], d.coef[l] = affs[l].data_ptr(), ds[l].data_ptr(), coefs[l].data_ptr()
            d.dgamma[l], d.dbeta[l] = dgs[l].data_ptr(), dbe[l].data_ptr()
        d.dout, d.dxa, d.dxb = dout.data_ptr(), _ptr(dxa), _ptr(dxb)
        need = lib.pf_bnmlp_train_ws_floats(ctypes.byref(d))
        ws = _ws(dev, need)
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
        d.stat = _stat(dev).data_ptr()
        sum_in = len(ctx.cfg) > 6 and ctx.cfg[6]
        d.flags = (2 if _DET else 0) | (4 if sum_in else 0)
        if len(ctx.cfg) > 5 and ctx.cfg[5]:
            _attach_sync(d, dev)
        _lib.check(lib.pf_bnmlp_train_bwd(ctypes.byref(d), _stream()), "pf_bnmlp_train_bwd")
        if sum_in:                                           # y[0] = xa + xb: both inputs take d[0]; layer 0 has no weights of its own
            return (ds[0], ds[0], None, None, None, dWs[1], dbs[1], dWs[2], dbs[2], dgs[0], dbe[0], dgs[1], dbe[1])
        return (dxa, dxb, None, dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2], dgs[0], dbe[0], dgs[1], dbe[1])


_NBT_PENDING = None        # inside forward_train: the BatchNorm layers whose batch counters are bumped by ONE launch at its end


def _count_batches(bns) -> None:
    """num_batches_tracked += 1 (torch.nn.BatchNorm in train mode): deferred to the end of the training forward when one is
    running (a launch per unit is pure latency on the step's dependency chain), immediate otherwise."""
    if _NBT_PENDING is not None:
        _NBT_PENDING.extend(bns)
        return
    with torch.no_grad():
        torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)


def bnmlp_fused(mlp, xa: Tensor, xb=None, last=None, sum_inputs: bool = False) -> Tensor:
    """last = (W, b): other tensors for the last (linear) layer - the next module's first layer folded in (interp_weights);
    sum_inputs: layer 0 = xa + xb (its weights live folded in the two producers' last layers)."""
    convs, bns = [mlp[0], mlp[3], mlp[6]], [mlp[1], mlp[4]]
    cfg = (0.01, float(bns[0].eps), float(bns[0].momentum), [bn.running_mean for bn in bns], [bn.running_var for bn in bns],
           _sync_bn_active(), bool(sum_inputs))
    prm = []
    for i, c in enumerate(convs):
        if i == 2 and last is not None:
            prm += [last[0], last[1]]
            continue
        if i == 0 and sum_inputs:                               # shapes only: no gradient comes back for these two
            prm += [c.weight.detach(), c.bias.detach()]
            continue
        prm += [c.weight, c.bias]
    for bn in bns:
        prm += [bn.weight, bn.bias]
    out = BnMlpFn.apply(xa, xb, cfg, *prm)
    _count_batches(bns)
    return out


_SIDE = {}


def _side_stream(dev, k: int = 0) -> "torch.cuda.Stream":
    st = _SIDE.get((dev, k))
    if st is None:
        st = _SIDE[(dev, k)] = torch.cuda.Stream(device=dev)
    return st


_DESC_BUF = {}


def _desc_buf(dev) -> Tensor:
    key = (dev, _stream())
    t = _DESC_BUF.get(key)
    if t is None:
        t = torch.empty(16 * ctypes.sizeof(_lib.PfMlpTrain), dtype=torch.uint8, device=dev)
        _DESC_BUF[key] = t
    return t


class CondNetBatchFn(Function):
    """Several LinearA1D conditioners (first layer without bias, interpflow.py:22-43) that only read conditioning features -
    the scale and shift nets of every flow block - in ONE launch forward and four backward (csrc/train_mlp.hip, batched
    entry points).  apply(cidx, *cs, *[W0, W1, b1, W2, b2 per net]) -> one [rows, dout] tensor per net; cidx[k] = which of
    the `cs` tensors net k reads."""

    @staticmethod
    def forward(ctx, cidx, *ts):
        lib = _lib.load()
        n = len(cidx)
        ncs = len(ts) - 5 * n
        cs = [c.contiguous() for c in ts[:ncs]]
        prm = [w.contiguous() for w in ts[ncs:]]
        descs = (_lib.PfMlpTrain * n)()
        outs, hs = [], []
        for k in range(n):
            W0, W1, b1, W2, b2 = prm[5 * k:5 * k + 5]
            d, rows = MlpFn._desc(None, cs[cidx[k]], 0, 1, (0.01, 0.01), [W0, W1, W2], [None, b1, b2])
            f32 = dict(dtype=torch.float32, device=W0.device)
            h = [torch.empty((rows, W0.shape[0]), **f32), torch.empty((rows, W1.shape[0]), **f32)]
            out = torch.empty((rows, W2.shape[0]), **f32)
            d.h[0], d.h[1], d.out = h[0].data_ptr(), h[1].data_ptr(), out.data_ptr()
            descs[k] = d
            outs.append(out)
            hs += h
        _lib.check(lib.pf_mlp_train_fwd_batch(descs, n, _desc_buf(cs[0].device).data_ptr(), _stream()), "pf_mlp_train_fwd_batch")
        ctx.cidx, ctx.ncs = cidx, ncs
        ctx.save_for_backward(*cs, *prm, *hs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        lib = _lib.load()
        cidx, ncs = ctx.cidx, ctx.ncs
        n = len(cidx)
        sv = list(ctx.saved_tensors)
        cs, prm, hs = sv[:ncs], sv[ncs:ncs + 5 * n], sv[ncs + 5 * n:]
        dev = cs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        descs = (_lib.PfMlpTrain * n)()
        keep, grads, dcs = [], [], [[] for _ in range(ncs)]
        need_tot = 0
        for k in range(n):
            W0, W1, b1, W2, b2 = prm[5 * k:5 * k + 5]
            d, rows = MlpFn._desc(None, cs[cidx[k]], 0, 1, (0.01, 0.01), [W0, W1, W2], [None, None, None])
            need_tot += lib.pf_mlp_train_ws_floats(ctypes.byref(d))
        ws = _ws(dev, need_tot)
        off = 0
        for k in range(n):
            W0, W1, b1, W2, b2 = prm[5 * k:5 * k + 5]
            c = cs[cidx[k]]
            d, rows = MlpFn._desc(None, c, 0, 1, (0.01, 0.01), [W0, W1, W2], [None, None, None])
            dout = douts[k].contiguous() if douts[k] is not None else torch.zeros((rows, W2.shape[0]), **f32)
            dz = [torch.empty_like(hs[2 * k]), torch.empty_like(hs[2 * k + 1])]
            dc = torch.empty_like(c)
            dW = [torch.empty_like(W0), torch.empty_like(W1), torch.empty_like(W2)]
            db = [torch.empty_like(b1), torch.empty_like(b2)]
            d.h[0], d.h[1], d.dz[0], d.dz[1] = hs[2 * k].data_ptr(), hs[2 * k + 1].data_ptr(), dz[0].data_ptr(), dz[1].data_ptr()
            d.dout, d.dc = dout.data_ptr(), dc.data_ptr()
            for l in range(3):
                d.dW[l] = dW[l].data_ptr()
            d.db[1], d.db[2] = db[0].data_ptr(), db[1].data_ptr()
            need = lib.pf_mlp_train_ws_floats(ctypes.byref(d))
            d.ws, d.ws_floats = ws.data_ptr() + 4 * off, need
            off += need
            descs[k] = d
            keep += [dout, dz]
            dcs[cidx[k]].append(dc)
            grads += [dW[0], dW[1], db[0], dW[2], db[1]]
        _lib.check(lib.pf_mlp_train_bwd_batch(descs, n, _desc_buf(dev).data_ptr(), _stream()), "pf_mlp_train_bwd_batch")
        dc_out = []
        for lst in dcs:
            if not lst:
                dc_out.append(None)
            elif len(lst) == 1:
                dc_out.append(lst[0])
            else:
                dc_out.append(torch.stack(lst).sum(0) if len(lst) > 2 else lst[0] + lst[1])
        return (None, *dc_out, *grads)


class MergeBatchFn(Function):
    """The FeatMergeUnits of all EdgeConv units (Linear + ReLU + Linear without bias, interpflow.py:251-258) in ONE launch forward
    and three backward (csrc/train_mlp.hip, batched entry points: one descriptor per unit, shapes may differ) instead of one /
    three per unit: their outputs are only read by the flow stage, so nothing waits for them before the last unit is done.
    apply(n, *hs, *[W1, b1, W2 per unit]) -> ONE flat tensor, the units' [rows, cdim] outputs one after the other (the layout the
    flow chains and the injector stack read: no concatenation afterwards).  Same kernels, same arithmetic as `mlp_fused`."""

    @staticmethod
    def _descs(hs, prm):
        n = len(hs)
        descs = (_lib.PfMlpTrain * n)()
        rows = None
        for k in range(n):
            W1, b1, W2 = prm[3 * k:3 * k + 3]
            descs[k], rows = MlpFn._desc(None, hs[k], 0, 1, (0.0,), [W1, W2], [b1, None])
        return descs, rows

    @staticmethod
    def forward(ctx, n, *ts):
        lib = _lib.load()
        hs = [h.contiguous() for h in ts[:n]]
        prm = [w.contiguous() for w in ts[n:]]
        descs, rows = MergeBatchFn._descs(hs, prm)
        f32 = dict(dtype=torch.float32, device=hs[0].device)
        mids = [torch.empty((rows, prm[3 * k].shape[0]), **f32) for k in range(n)]
        cds = [int(prm[3 * k + 2].shape[0]) for k in range(n)]
        flat = torch.empty((rows * sum(cds),), **f32)
        off = 0
        for k in range(n):
            descs[k].h[0], descs[k].out = mids[k].data_ptr(), flat.data_ptr() + 4 * off
            off += rows * cds[k]
        _lib.check(lib.pf_mlp_train_fwd_batch(descs, n, _desc_buf(hs[0].device).data_ptr(), _stream()), "pf_mlp_train_fwd_batch")
        ctx.n = n
        ctx.save_for_backward(*hs, *prm, *mids)
        return flat

    @staticmethod
    def backward(ctx, dflat):
        lib = _lib.load()
        n = ctx.n
        dflat = dflat.contiguous()
        sv = list(ctx.saved_tensors)
        hs, prm, mids = sv[:n], sv[n:4 * n], sv[4 * n:]
        dev = hs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        descs, rows = MergeBatchFn._descs(hs, prm)
        need = [lib.pf_mlp_train_ws_floats(ctypes.byref(descs[k])) for k in range(n)]
        ws = _ws(dev, sum(need))
        keep, dhs, grads, off, doff = [], [], [], 0, 0
        for k in range(n):
            W1, b1, W2 = prm[3 * k:3 * k + 3]
            dout = dflat[doff:doff + rows * W2.shape[0]]
            doff += rows * W2.shape[0]
            dz, dh = torch.empty_like(mids[k]), torch.empty_like(hs[k])
            dW1, db1, dW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
            d = descs[k]
            d.h[0], d.dz[0], d.dout, d.dc = mids[k].data_ptr(), dz.data_ptr(), dout.data_ptr(), dh.data_ptr()
            d.dW[0], d.dW[1], d.db[0] = dW1.data_ptr(), dW2.data_ptr(), db1.data_ptr()
            d.ws, d.ws_floats = ws.data_ptr() + 4 * off, need[k]
            off += need[k]
            descs[k] = d
            keep += [dout, dz]
            dhs.append(dh)
            grads += [dW1, db1, dW2]
        _lib.check(lib.pf_mlp_train_bwd_batch(descs, n, _desc_buf(dev).data_ptr(), _stream()), "pf_mlp_train_bwd_batch")
        return (None, *dhs, *grads)


def mlp_fused(y, c: Tensor, td: int, cdiv: int, slopes, layers) -> Tensor:
    """layers: nn.Linear modules.  -> [rows, out]"""
    wb = []
    for lin in layers:
        wb += [lin.weight, lin.bias]
    return MlpFn.apply(y, c, td, cdiv, tuple(slopes), *wb)


_FUSED = os.environ.get("PF_TRAIN_FUSED", "1") != "0"
_MERGE_BATCH = os.environ.get("PF_TRAIN_MERGE_BATCH", "1") != "0"   # the FeatMergeUnits of all units in one batched launch; "0" = one per unit
_FOLD_WU = os.environ.get("PF_TRAIN_FOLD_WU", "1") != "0"    # the weight unit's first conv folded into its producers (interp_weights); "0" = A/B reference


def knn_csr(idx: Tensor):
    """Transposed neighbour lists of idx [B,N,K] int32 (batch-local): (off [T+1], edge [T*K]) - for every point the edges that
    point AT it.  Built once per step (4 small launches) and shared by all EdgeConv units on the same idx: their backward then
    gathers dQ instead of scatter-adding it with float atomics."""
    B, N, K = idx.shape
    T = B * N
    dev = idx.device
    off = torch.empty(T + 1, dtype=torch.int32, device=dev)
    edge = torch.empty(T * K, dtype=torch.int32, device=dev)
    cnt = torch.empty((T + 3) // 4 * 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().pf_knn_csr(idx.data_ptr(), B, N, K, off.data_ptr(), edge.data_ptr(), cnt.data_ptr(), _stream()),
               "pf_knn_csr")
    if _DET:                                               # one summation order over every list, run after run
        _lib.check(_lib.load().pf_knn_csr_sort(off.data_ptr(), edge.data_ptr(), T, _stream()), "pf_knn_csr_sort")
    return off, edge


def knn_csr_pair(idx: Tensor, K2: int):
    """knn_csr(idx) and knn_csr(idx[..., :K2].contiguous()) from one pass over idx (pf_knn_csr_pair: 4 launches instead of 8)."""
    B, N, K = idx.shape
    T = B * N
    dev = idx.device
    i32 = dict(dtype=torch.int32, device=dev)
    off, edge = torch.empty(T + 1, **i32), torch.empty(T * K, **i32)
    off2, edge2 = torch.empty(T + 1, **i32), torch.empty(T * K2, **i32)
    cnt = torch.empty(2 * ((T + 3) // 4 * 4), **i32)
    lib = _lib.load()
    _lib.check(lib.pf_knn_csr_pair(idx.data_ptr(), B, N, K, K2, off.data_ptr(), edge.data_ptr(), off2.data_ptr(), edge2.data_ptr(),
                                   cnt.data_ptr(), _stream()), "pf_knn_csr_pair")
    if _DET:
        _lib.check(lib.pf_knn_csr_sort(off.data_ptr(), edge.data_ptr(), T, _stream()), "pf_knn_csr_sort")
        _lib.check(lib.pf_knn_csr_sort(off2.data_ptr(), edge2.data_ptr(), T, _stream()), "pf_knn_csr_sort")
    return (off, edge), (off2, edge2)


def _ec_fused_supported(p, x: Tensor, idx: Tensor, pooling: bool) -> bool:
    """Shapes the fused unit kernels are built for (csrc/train_fused.hip: ec_dims); anything else takes the per-op path."""
    g, nconv, odim = p.convs[0][0].weight.shape[0], len(p.convs), p.conv_out.weight.shape[0]
    B, N, _ = x.shape
    K = idx.shape[-1]
    return (g in (8, 16, 32) and 1 <= nconv <= 8 and g * nconv in (32, 64, 128) and odim % 16 == 0 and 16 <= odim <= 128
            and (B * N * K) % 16 == 0 and (K == 16 or not pooling))


# The interpolation branch runs on the side stream either way; this is only WHEN the host issues it: before the feature units
# (default) or after them - autograd replays nodes in reverse creation order, so issued late it is the FIRST thing the backward
# enqueues after the flow stage instead of the last
_SIDE_LATE = os.environ.get("PF_TRAIN_SIDE_LATE", "0") == "1"
_SIDE_STREAM = os.environ.get("PF_TRAIN_STREAMS", "1") != "0"      # "0": the interpolation branch on the calling stream (timing reference)
_TAP = os.environ.get("PF_TRAIN_TAP", "1") != "0"     # a unit's output gradient from its merge unit added inside the next unit's dx GEMM
_PREFOLD = os.environ.get("PF_TRAIN_PREFOLD", "1") != "0"     # the feature units' folded weights in one launch at the top of the forward


def ec_prefold(units, x0: Tensor, K: int):
    """Folded edge-feature weights (Wpq, bpq) of a CHAIN of pooled units (unit i + 1 reads unit i's output) in one launch
    (pf_ec_train_fold_batch) -> one pair per unit for `edgeconv_train_fused(..., prefold=)`, or None where the fused path
    does not apply.  Parameters only: valid until the next optimizer update."""
    if not (_FUSED and _PREFOLD) or _UNFOLDED or not 1 <= len(units) <= 8:
        return None
    B, N, C = x0.shape
    if (B * N * K) % 16:
        return None
    dev = x0.device
    f32 = dict(dtype=torch.float32, device=dev)
    descs = (_lib.PfEcTrain * len(units))()
    outs = []
    for k, p in enumerate(units):
        convs = [seq[0] for seq in p.convs] + [p.conv_out]
        g, nconv, odim = convs[0].weight.shape[0], len(p.convs), p.conv_out.weight.shape[0]
        if g not in (8, 16, 32) or g * nconv not in (32, 64, 128) or odim % 16 or not 16 <= odim <= 128 or K != 16:
            return None
        if any(c.weight.dtype != torch.float32 or not c.weight.is_contiguous() or c.bias is None for c in convs):
            return None
        if convs[0].weight.reshape(g, -1).shape[1] != 3 * C:
            return None
        S = g * nconv + odim
        Wpq, bpq = torch.empty((2 * S, C), **f32), torch.empty((2 * S,), **f32)
        d = _lib.PfEcTrain()
        d.B, d.N, d.K, d.C, d.growth, d.nconv, d.odim, d.pooling = B, N, K, C, g, nconv, odim, 1
        for t, c in enumerate(convs):
            d.W[t], d.bias[t] = c.weight.data_ptr(), c.bias.data_ptr()
        d.Wpq, d.bpq = Wpq.data_ptr(), bpq.data_ptr()
        descs[k] = d
        outs.append((Wpq, bpq))
        C = odim
    _lib.check(_lib.load().pf_ec_train_fold_batch(descs, len(units), _stream()), "pf_ec_train_fold_batch")
    return outs


def edgeconv_train_fused(p, x: Tensor, idx: Tensor, pooling: bool = True, csr=None, persistent: bool = False, out=None,
                         prefold=None, tap: bool = False) -> Tensor:
    """out = (W, b): other tensors for conv_out (the next module's first layer folded in, interp_weights).
    prefold = (Wpq, bpq) from `ec_prefold`.  tap: return (result, x) - an alias of x for x's other consumer, whose gradient the
    unit's backward then adds inside its dx GEMM (EdgeConvUnitFn)."""
    convs = [seq[0] for seq in p.convs] + [p.conv_out]
    bns = [seq[1] for seq in p.convs]
    g, nconv, odim = convs[0].weight.shape[0], len(bns), p.conv_out.weight.shape[0]
    if out is not None:
        ws = [c.weight for c in convs[:-1]] + [out[0]]
        bs = [c.bias for c in convs[:-1]] + [out[1]]
        cfg = (idx.shape[-1], g, nconv, odim, bool(pooling), 0.05, float(bns[0].eps), float(bns[0].momentum),
               [bn.running_mean for bn in bns], [bn.running_var for bn in bns], csr,
               bool(persistent) and _PERSIST and not _sync_bn_active(), _sync_bn_active(),
               True)         # [13]: conv_out's gradient is consumed INSIDE the pass (FoldWuFn): no weight-gradient stream for this unit
        res = EdgeConvUnitFn.apply(x, idx, cfg, *ws, *bs, *[bn.weight for bn in bns], *[bn.bias for bn in bns])
        _count_batches(bns)
        return res
    cfg = (idx.shape[-1], g, nconv, odim, bool(pooling), 0.05, float(bns[0].eps), float(bns[0].momentum),
           [bn.running_mean for bn in bns], [bn.running_var for bn in bns], csr,
           bool(persistent) and _PERSIST and not _sync_bn_active(), _sync_bn_active(), False, prefold,
           bool(tap) and x.requires_grad)
    out = EdgeConvUnitFn.apply(x, idx, cfg, *[c.weight for c in convs], *[c.bias for c in convs],
                               *[bn.weight for bn in bns], *[bn.bias for bn in bns])
    _count_batches(bns)
    return out


def cond_net(net, h: Tensor) -> Tensor:
    """LinearA1D (interpflow.py:38-43)."""
    L = net.layers
    h = ActFn.apply(linear(h, L[0].weight), 0.01)
    h = ActFn.apply(linear(h, L[2].weight, L[2].bias), 0.01)
    return linear(h, L[4].weight, L[4].bias)


def cond_net_fused(net, y, c: Tensor, td: int, cdiv: int) -> Tensor:
    """LinearA1D on cat[y[..., :td], c[row // cdiv]] in one launch (csrc/train_mlp.hip); -> [rows, dout].
    net: the LinearA1D module, or its three linear layers (weight / bias holders) as a list."""
    L = net if isinstance(net, (list, tuple)) else [net.layers[0], net.layers[2], net.layers[4]]
    if cdiv not in (1, 2, 4, 8, 16):                   # the kernel shares a conditioning row between 2^k replicas only
        c, cdiv = RepeatRowsFn.apply(c, cdiv), 1
    return mlp_fused(y, c, td, cdiv, (0.01, 0.01), L)


def cond_net_split(net, h1: Tensor, cpart: Tensor) -> Tensor:
    """LinearA1D on cat[h1, c] with the c-columns of the bias-free first layer already applied: W0 [h1; c] = W0[:, :td] h1 +
    cpart (interpflow.py:38-41).  cpart = W0[:, td:] c is per ORIGINAL point: f and the R replicas of g share one evaluation."""
    L = net.layers
    td = h1.shape[-1]
    h = ActFn.apply(linear(h1, L[0].weight[:, :td]) + cpart, 0.01)
    h = ActFn.apply(linear(h, L[2].weight, L[2].bias), 0.01)
    return linear(h, L[4].weight, L[4].bias)


def _mlp_bn(mlp, x: Tensor) -> Tensor:
    """Conv,BN,LReLU(.01),Conv,BN,LReLU,Conv on rows (DistanceEncoder / WeightEstimationUnit)."""
    x = bn_lrelu(linear(x, mlp[0].weight, mlp[0].bias), mlp[1], 0.01)
    x = bn_lrelu(linear(x, mlp[3].weight, mlp[3].bias), mlp[4], 0.01)
    return linear(x, mlp[6].weight, mlp[6].bias)


class ParamFanFn(torch.autograd.Function):
    """Identity on parameters that are used several times in one forward: `uses[i]` aliases of params[i].  Autograd sums the
    gradients of a tensor's uses with one small add launch per extra use - 54 of them per step for the flow blocks'
    parameters (ActNorm, W, the coupling net: f and g share them).  Here the sums of ALL parameters are two multi-tensor
    launches in this node's backward."""

    @staticmethod
    def forward(ctx, uses, *params):
        ctx.uses = uses
        outs = []
        for p, u in zip(params, uses):
            outs += [p.view_as(p) for _ in range(u)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        groups, k = [], 0
        for u in ctx.uses:
            groups.append([g for g in grads[k:k + u] if g is not None])
            k += u
        acc = [gs[0] if gs else None for gs in groups]
        for level in range(1, max(ctx.uses)):
            sel = [i for i, gs in enumerate(groups) if len(gs) > level]
            if sel:
                summed = torch._foreach_add([acc[i] for i in sel], [groups[i][level] for i in sel])
                for i, t in zip(sel, summed):
                    acc[i] = t
        return (None, *acc)


class _Lin:
    """weight / bias holder with the attribute names of nn.Linear (for mlp_fused on parameter aliases)."""
    __slots__ = ("weight", "bias")

    def __init__(self, weight, bias=None):
        self.weight, self.bias = weight, bias


_FAN = os.environ.get("PF_TRAIN_FAN", "1") != "0"
# all flow blocks of a direction as one autograd node (FlowChainFn); "0" = one node per block piece (the A/B reference)
_CHAIN = os.environ.get("PF_TRAIN_CHAIN", "1") != "0"
# chains of one-element torch launches as fused kernels (csrc/train_glue.hip: interpolation of the latent, the log-likelihood
# inside the f chain, the loss head of loss.PuganLossFn); "0" = the torch expressions (the A/B reference)
_GLUE = os.environ.get("PF_TRAIN_GLUE", "1") != "0"
_CSR_SIDE = os.environ.get("PF_TRAIN_CSR_SIDE", "1") != "0"      # A/B switch, read once at import like its neighbours (README "Switches")


def _flow_param_aliases(net):
    """Per flow block: aliases of the parameters f and g share (ParamFanFn) - dict of lists indexed by use."""
    plist = []
    for blk in net.flow_blocks:
        L = blk.coupling1.bias_net.layers
        plist += [(blk.actnorm.logs, 3), (blk.actnorm.bias, 2), (blk.permutate1.permutater.W, 2), (L[0].weight, 2),
                  (L[2].weight, 2), (L[2].bias, 2), (L[4].weight, 2), (L[4].bias, 2)]
    if _FAN:
        flat = list(ParamFanFn.apply(tuple(u for _, u in plist), *[p for p, _ in plist]))
    else:
        flat = [p for p, u in plist for _ in range(u)]
    out, k = [], 0
    for _ in net.flow_blocks:
        a = {}
        for name, u in (("logs", 3), ("bias", 2), ("W", 2), ("w0", 2), ("w2", 2), ("b2", 2), ("w4", 2), ("b4", 2)):
            a[name] = flat[k:k + u]
            k += u
        a["net"] = [[_Lin(a["w0"][j]), _Lin(a["w2"][j], a["b2"][j]), _Lin(a["w4"][j], a["b4"][j])] for j in range(2)]
        out.append(a)
    return out


def _flow_chain_params(net):
    """The eight parameters of every flow block, twice (f and g share them): two alias lists for FlowChainFn."""
    plist = []
    for blk in net.flow_blocks:
        L = blk.coupling1.bias_net.layers
        plist += [blk.actnorm.logs, blk.actnorm.bias, blk.permutate1.permutater.W, L[0].weight, L[2].weight, L[2].bias,
                  L[4].weight, L[4].bias]
    if _FAN:
        flat = list(ParamFanFn.apply(tuple(2 for _ in plist), *plist))
        return flat[0::2], flat[1::2]
    return plist, plist


def forward_train(net, xyz: Tensor, upratio: int) -> Tuple[Tensor, Tensor]:
    """PointInterpFlow.forward in train() mode (interpflow.py:327-337) with gradients."""
    global _NBT_PENDING
    _NBT_PENDING = []
    set_deterministic(getattr(net, "deterministic", False))
    global _DW_NET
    _DW_NET = bool(getattr(net, "train_dw_stream", False))
    _FC_SCOPE[1] += 1
    _FC_SCOPE[0] = _FC_SCOPE[1]
    try:
        with sync_bn(getattr(net, "sync_batchnorm", False)):
            return _forward_train(net, xyz, upratio)
    finally:
        _FC_SCOPE[0] = 0
        _FC_IMG.clear()
        pending, _NBT_PENDING = _NBT_PENDING, None
        if pending:                                   # on the calling stream, after the side stream has been joined
            with torch.no_grad():
                torch._foreach_add_([bn.num_batches_tracked for bn in pending], 1)


def _forward_train(net, xyz: Tensor, upratio: int) -> Tuple[Tensor, Tensor]:
    from . import ops
    xyz = xyz.detach().contiguous().float()
    B, N, _ = xyz.shape
    R = upratio
    idx16, _ = ops.knn_idx32(xyz, xyz, 16)
    fused_ec = _FUSED                                     # also under SyncBN: the fused kernels defer a layer's statistics to an all-reduce
    use_side = getattr(net, "train_streams", True) and not _sync_bn_active() and _SIDE_STREAM
    # the 8 nearest of the 16: only the interpolation branch (and, after the join, the latent's interpolation) reads them - with a
    # side stream the copy is made there, off the chain knn -> fold -> first unit
    idx8 = None if use_side else idx16[..., :8].contiguous()
    # transposed neighbour lists: only the BACKWARD of the EdgeConv units reads them - with a side stream they are built there,
    # off the main chain (8 small launches, ~55 us), and joined with the interpolation weights
    csr16 = csr8 = None
    csr_side = use_side and _CSR_SIDE
    if fused_ec and not csr_side:
        csr16, csr8 = knn_csr_pair(idx16, 8)

    # ---- interpolation weights (interpflow.py:85-151): a function of xyz and the neighbour lists only, independent of the
    # feature extractor / flow f chain that follows.  At 32 x 256 points no kernel of the step fills the chip, so this branch
    # (two BatchNorm MLPs + one EdgeConv unit, forward and - autograd keeps the stream - backward) runs on a side stream
    # beside the main chain; inside a captured step it becomes a parallel branch of the graph.  Scratch buffers are per
    # (device, stream) (_ws / _stat), results are identical to the one-stream order (net.train_streams = False).
    def interp_weights():
        ip = net.interp
        fd = torch.empty((B * N * 8, 10), dtype=torch.float32, device=xyz.device)        # inputs only: no gradient
        _lib.check(_lib.load().pf_dist_feature(xyz.data_ptr(), idx8.data_ptr(), B, N, 8, fd.data_ptr(), _stream()), "pf_dist_feature")
        fused_bn = _FUSED
        de, wu, fc = ip.knn_context.distance_encoder.mlp, ip.weight_unit.mlp, ip.knn_context.feat_conv
        if (fused_bn and _FOLD_WU and _ec_fused_supported(fc, xyz, idx8, False)
                and wu[0].weight.shape[0] == de[6].weight.shape[0] == fc.conv_out.weight.shape[0] == wu[0].weight.shape[1] // 2):
            # The weight unit's first conv sees cat[d, feat] with NO nonlinearity after the producers' last (linear) layers
            # (interpflow.py:134,144-146), so W0 [d; feat] + b0 = (W0a W6) a2 + (W0b Wout) e + (W0a b6 + W0b bout + b0): the
            # producers' last layers run with the PRODUCT weights and emit the two halves of the pre-BatchNorm output directly;
            # the first layer of the weight unit is their sum (PF_BNMLP_SUM_INPUTS) - two [E8, 128] x [128, 128] products per
            # direction less, same function.  The products and their chain rule back to W0, W6, Wout and the three biases are one
            # small launch each (FoldWuFn; the same algebra as the eval path's packing.fold_state_dict).
            W6f, b6f, Wof, bof = FoldWuFn.apply(wu[0].weight, wu[0].bias, de[6].weight, de[6].bias, fc.conv_out.weight,
                                                fc.conv_out.bias)
            d = bnmlp_fused(de, fd, last=(W6f, b6f))
            feat = edgeconv_train_fused(fc, xyz, idx8, False, csr8, False, out=(Wof, bof))
            return bnmlp_fused(wu, d, feat, sum_inputs=True)
        d = bnmlp_fused(de, fd) if fused_bn else _mlp_bn(de, fd)
        feat = edgeconv_train(fc, xyz, idx8, pooling=False, csr=csr8)                            # d, feat: [E8,128]
        if fused_bn:
            return bnmlp_fused(wu, d, feat)                       # on cat[d, feat] (interpflow.py:146) without building it
        return _mlp_bn(ip.weight_unit.mlp, torch.cat([d, feat], dim=1))  # [E8,32]

    # folded edge-feature weights of the six feature units: parameters only - one launch here instead of one per unit
    pre = ec_prefold(list(net.feat_convs), xyz, 16) if (fused_ec and not _sync_bn_active()) else None

    side = None
    if use_side:
        side = _side_stream(xyz.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            idx8 = idx16[..., :8].contiguous()
            if fused_ec and csr_side:
                csr16, csr8 = knn_csr_pair(idx16, 8)
            if not _SIDE_LATE:
                w = interp_weights()
    else:
        w = interp_weights()

    # ---- feature extractor  (the merge units on a second side stream beside EdgeConv unit i + 1 were tried: 7.65 -> 8.35 ms per
    # captured step - six fork / join pairs of tiny kernels cost more in graph dependencies than their overlap returns)
    cs: List[Tensor] = []
    hs_all: List[Tensor] = []
    h = xyz
    for i in range(net.num_blocks):
        # the main chain's units may run as persistent grid-barrier launches: nothing else with a grid barrier runs beside them
        # (the side stream's interpolation unit keeps the per-layer kernels; the EMD auction starts after the forward)
        # An explicit opt-in: TrainerModule sets `network.train_persistent` from its configuration (on, unless the
        # device is shared: cfg.emd_workgroups == 1 / cfg.persistent_kernels = False); a bare PointInterpFlow.train() forward on a
        # GPU it may share with another process keeps the per-layer kernels - a grid barrier whose workgroups are not all resident
        # spins for seconds before its bounded time-out makes the output NaN.
        batch_merge = _FUSED and _MERGE_BATCH and net.num_blocks <= 16
        h = edgeconv_train(net.feat_convs[i], h, idx16, csr=csr16, persistent=getattr(net, "train_persistent", False),
                           prefold=pre[i] if pre is not None else None, tap=_TAP and batch_merge and i > 0)
        if isinstance(h, tuple):                                  # (output, the input again): the merge unit reads the alias, so
            h, hs_all[i - 1] = h                                  # its gradient reaches the previous unit through THIS unit's dx GEMM
        m = net.merge_convs[i]
        if _FUSED and _MERGE_BATCH and net.num_blocks <= 16:
            hs_all.append(h)                                      # all merge units in one batched launch below
        elif _FUSED:
            cs.append(mlp_fused(None, h, 0, 1, (0.0,), [m.conv1, m.conv2]).view(B, N, -1))
        else:
            cs.append(linear(ActFn.apply(linear(h, m.conv1.weight, m.conv1.bias), 0.0), m.conv2.weight))
    if side is not None and _SIDE_LATE:
        with torch.cuda.stream(side):
            w = interp_weights()
    if hs_all:
        mp = []
        for m in net.merge_convs:
            mp += [m.conv1.weight, m.conv1.bias, m.conv2.weight]
        cflat_m = MergeBatchFn.apply(len(hs_all), *hs_all, *mp)
        off = 0
        for m in net.merge_convs:
            cd = m.conv2.weight.shape[0]
            cs.append(cflat_m[off:off + B * N * cd].view(B, N, cd))
            off += B * N * cd

    # ---- injector nets (s, t) of every block: functions of cs[i] only - one batched launch (and shared by f and g)
    st_all = None
    st_prm = []
    nb = net.num_blocks
    chain = (_CHAIN and _FUSED and nb <= 8 and R in (1, 2, 4, 8, 16)
             and all(b.actnorm.is_inited for b in net.flow_blocks) and all(c.shape[-1] in (32, 64, 128) for c in cs)
             and all(b.coupling1.bias_net.layers[2].weight.shape == (64, 64) for b in net.flow_blocks))
    if _FUSED:
        for blk in net.flow_blocks:
            for cn in (blk.coupling2.scale_net, blk.coupling2.bias_net):
                L = cn.layers
                st_prm += [L[0].weight, L[2].weight, L[2].bias, L[4].weight, L[4].bias]
        if not chain:
            cidx = tuple(i for i in range(net.num_blocks) for _ in range(2))
            st_all = CondNetBatchFn.apply(cidx, *cs, *st_prm)

    # ---- f, interpolation, g with every flow block of a direction in one autograd node (csrc/train_flowchain.hip).  ActNorm's
    # data-dependent init needs each block's input on the host side of the chain: the first step takes the per-block path below
    if chain:
        pf, pg = _flow_chain_params(net)
        ccs = tuple(int(c.shape[-1]) for c in cs)
        cflat = cflat_m if hs_all else torch.cat([c.reshape(-1) for c in cs])       # one tensor for its consumers
        cflat_t = None
        cflat_f = cflat_g = cflat
        if _FANOUT:                                                   # their four gradients: one launch (FanoutFn)
            cflat, cflat_t, cflat_f, cflat_g = FanoutFn.apply(4, cflat)
        st = CondNetStackFn.apply(ccs, B * N, cflat, cflat_t, *st_prm)
        if _GLUE and nb < 8:
            z, logp1 = FlowChainFn.apply(0, 1, float(N), ccs, xyz, cflat_f, st, B, *pf)     # log-likelihood from the kernel's epilogue
            logp = logp1.view(())
        else:
            z, ssum, ld = FlowChainFn.apply(0, 1, float(N), ccs, xyz, cflat_f, st, 0, *pf)
            logp = -(BatchSumFn.apply(z, 1).mean() + ld.sum() - ssum.sum() / B)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            w.record_stream(torch.cuda.current_stream())
            for t in (csr16 or ()) + (csr8 or ()) + (idx8,):
                t.record_stream(torch.cuda.current_stream())
        if _GLUE and R <= 8:
            u = InterpWsumFn.apply(w.view(B * N, 8, -1), z, idx8, R, csr8 if _DET else None)
        else:
            zj = GatherRowsFn.apply(z, idx8)
            fz = SoftmaxWsumFn.apply(w.view(B * N, 8, -1), zj.view(B * N, 8, 3), R)
            u = fz.transpose(1, 2).reshape(B, N * R, 3)
        x = FlowChainFn.apply(1, R, float(N), ccs, u, cflat_g, st, 0, *pg)
        return x, logp

    # ---- f + log-likelihood
    p = xyz
    ldj = torch.zeros(B, device=xyz.device)
    st_nets: List[Tuple[Tensor, Tensor]] = []                      # injector (s, t) per block: functions of cs[i] only, shared by f and g
    cparts: List[Tensor] = []                                      # c-columns of coupling1's first layer, likewise
    winvs: List[Tensor] = []
    lds: List[Tensor] = []
    ssums: List[Tensor] = []
    alias = _flow_param_aliases(net) if _FUSED else None          # aliases share the parameters' storage (ActNorm init below)
    for i in range(net.num_blocks):
        blk = net.flow_blocks[i]
        an = blk.actnorm
        if not an.is_inited:                                       # normalize.py:45-54
            with torch.no_grad():
                an.bias.data.copy_(-torch.mean(p.detach(), dim=(0, 1), keepdim=True))
                an.logs.data.copy_(-torch.log(torch.std(p.detach(), dim=(0, 1), keepdim=True) + 1e-6))
                an.is_inited = True
        W = blk.permutate1.permutater.W
        td = 1 if i % 2 == 0 else 2
        if _FUSED:
            al = alias[i]
            Winv, ld = FlowParamsFn.apply(al["W"][0], al["logs"][0], float(N))    # W^-1 and (sum(logs) + log|det W|) N (permutate.py:119)
            winvs.append(Winv)
            lds.append(ld)
            y = FlowAffineFn.apply(p, None, 0, al["logs"][1], al["bias"][0], al["W"][1], 0)    # ActNorm + einsum 'ij,bnj->bni' (permutate.py:118)
            o = cond_net_fused(al["net"][0], y, cs[i], td, 1).view(B, N, -1)
            s, t = st_all[2 * i].view(B, N, -1), st_all[2 * i + 1].view(B, N, -1)
            st_nets.append((s, t))
            p, ssum = CoupleInject2Fn.apply(y, o, s, t, td)
            ssums.append(ssum)
            continue
        y = linear(ActNormFn.apply(p, an.logs, an.bias, 0), W)     # einsum 'ij,bnj->bni' (permutate.py:118)
        ld = (torch.sum(an.logs) + torch.log(torch.abs(_det_inv3(W)[0]))) * N        # parameter-only scalars (permutate.py:119)
        cparts.append(linear(cs[i], blk.coupling1.bias_net.layers[0].weight[:, td:]))
        o = cond_net_split(blk.coupling1.bias_net, y[..., :td], cparts[i])
        s = cond_net(blk.coupling2.scale_net, cs[i])
        t = cond_net(blk.coupling2.bias_net, cs[i])
        st_nets.append((s, t))
        p = CoupleInjectFn.apply(y, o, s, t, td)
        ldj = ldj + ld - BatchSumFn.apply(s, 0)
    z = p
    if _FUSED:
        # -mean_b(gauss_b + sum_i (ld_i - sum(s_i)[b])) with the batch mean taken once, over scalars: the same number as the
        # reference's per-sample bookkeeping (interpflow.py:327-337, probs.py:73-93)
        logp = -(BatchSumFn.apply(z, 1).mean() + torch.cat(lds).sum() - torch.cat(ssums).sum() / B)
    else:
        logp = -torch.mean(BatchSumFn.apply(z, 1) + ldj)

    # ---- interpolation: the weights w were started on the side stream before the feature extractor
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
        w.record_stream(torch.cuda.current_stream())
        for t in (csr16 or ()) + (csr8 or ()) + (idx8,):
            t.record_stream(torch.cuda.current_stream())
    zj = GatherRowsFn.apply(z, idx8)                              # [E8,3]
    fz = SoftmaxWsumFn.apply(w.view(B * N, 8, -1), zj.view(B * N, 8, 3), R)      # [T,3,R]
    u = fz.transpose(1, 2).reshape(B, N * R, 3)

    # ---- g (exact inverse); injector nets are evaluated per ORIGINAL point and replicated
    for i in reversed(range(net.num_blocks)):
        blk = net.flow_blocks[i]
        td = 1 if i % 2 == 0 else 2
        if _FUSED:
            v = InjectInv2Fn.apply(u, st_nets[i][0], st_nets[i][1], R)        # s, t of the original point: row // R
            al = alias[i]
            o = cond_net_fused(al["net"][1], v, cs[i], td, R).view(B, N * R, -1)
            u = FlowAffineFn.apply(v, o, td, al["logs"][2], al["bias"][1], winvs[i], 1)    # permutate.py:123-124
            continue
        s = RepeatRowsFn.apply(st_nets[i][0], R)                   # same nets, same input as in f: evaluated once (autograd sums both uses)
        t = RepeatRowsFn.apply(st_nets[i][1], R)
        v = InjectInvFn.apply(u, s, t)
        o = cond_net_split(blk.coupling1.bias_net, v[..., :td], RepeatRowsFn.apply(cparts[i], R))
        W = blk.permutate1.permutater.W
        u = linear(CoupleAddFn.apply(v, o, td), _det_inv3(W)[1])   # permutate.py:123-124 (3x3 inverse: parameter-only)
        u = ActNormFn.apply(u, blk.actnorm.logs, blk.actnorm.bias, 1)
    return u, logp
