"""PointInterpFlow on MI355X: the reference's module surface over the HIP kernels.

Drop-in for `modules/discrete/interpflow.py:262-350` (reference): same constructor, same
`forward / sample / feat_extract / f / g / log_prob / set_to_initialized_state` methods and the
same 408-entry `state_dict` (reference checkpoints load unchanged, ours load into the reference).
The sub-modules below only HOLD parameters under the reference's names; all arithmetic of the
inference path runs in libpuflow_hip.so (no torch math, no CPU fallback).

Weights are folded + packed once (puflow_amd/packing.py) and cached on the device; the cache is
dropped by load_state_dict(), .to()/.cuda() and train().  In train() mode forward() is differentiable
and runs the un-fused HIP training ops of puflow_amd/train_ops.py (BatchNorm batch statistics, ActNorm
first-batch init) - see DESIGN.md section 7 for the HIP / torch split of that path.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .packing import COND_CHANNELS, FEAT_CHANNELS, GROWTH, NUM_BLOCKS, fold_state_dict, pack_plan

_EC_CFG = [0, 1, 2, 2, 2, 2]
_CHECK_FINITE = os.environ.get("PF_CHECK_FINITE", "0") == "1"      # debug aid: verify every eval forward is finite (syncs)


# ----------------------------------------------------------------------------------------
# parameter holders (names = reference state_dict keys)
# ----------------------------------------------------------------------------------------
def _conv_bn(cin: int, cout: int, slope: float) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1), nn.BatchNorm2d(cout), nn.LeakyReLU(slope))


class _EdgeConvParams(nn.Module):
    """FeatureExtractUnit parameters (interpflow.py:190-221)."""

    def __init__(self, cin: int, odim: int, growth: int):
        super().__init__()
        nconv = odim // growth
        self.convs = nn.ModuleList([_conv_bn(3 * cin + growth * t, growth, 0.05) for t in range(nconv)])
        self.conv_out = nn.Conv2d(3 * cin + growth * nconv, odim, kernel_size=1)


class _MergeParams(nn.Module):
    """FeatMergeUnit parameters (interpflow.py:251-255)."""

    def __init__(self, idim: int, odim: int):
        super().__init__()
        self.conv1 = nn.Linear(idim, idim // 2, bias=True)
        self.conv2 = nn.Linear(idim // 2, odim, bias=False)


class _CondNet(nn.Module):
    """LinearA1D parameters (interpflow.py:22-36); last layer zero-initialised like the reference."""

    def __init__(self, din: int, dh: int, dout: int):
        super().__init__()
        last = nn.Linear(dh, dout, bias=True)
        nn.init.zeros_(last.weight)
        nn.init.zeros_(last.bias)
        self.layers = nn.Sequential(nn.Linear(din, dh, bias=False), nn.LeakyReLU(), nn.Linear(dh, dh, bias=True),
                                    nn.LeakyReLU(), last)


class _ActNormParams(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.logs = nn.Parameter(torch.zeros(1, 1, ch))
        self.bias = nn.Parameter(torch.zeros(1, 1, ch))
        self.is_inited = False           # plain attribute, not in the state_dict (normalize.py:28)


class _Holder(nn.Module):
    pass


class _FlowBlockParams(nn.Module):
    """FlowBlock parameters (interpflow.py:46-63)."""

    def __init__(self, idim: int, hdim: int, cdim: int, is_even: bool):
        super().__init__()
        self.actnorm = _ActNormParams(idim)
        self.permutate1 = _Holder()
        self.permutate1.permutater = _Holder()
        q, _ = np.linalg.qr(np.random.randn(idim, idim))          # permutate.py:102-105
        self.permutate1.permutater.W = nn.Parameter(torch.from_numpy(q.astype(np.float32)))
        self.permutate2 = _Holder()
        self.permutate2.permutater = _Holder()
        rev = torch.arange(idim - 1, -1, -1, dtype=torch.int64)
        self.permutate2.permutater.register_buffer("direct_idx", rev.clone())
        self.permutate2.permutater.register_buffer("inverse_idx", rev.clone())
        tdim = 1 if is_even else 2
        self.coupling1 = _Holder()
        self.coupling1.bias_net = _CondNet(tdim + cdim, hdim, idim - tdim)
        self.coupling2 = _Holder()
        self.coupling2.bias_net = _CondNet(cdim, hdim, idim)
        self.coupling2.scale_net = _CondNet(cdim, hdim, idim)


class _InterpParams(nn.Module):
    """InterpolationModule parameters (interpflow.py:85-151)."""

    def __init__(self):
        super().__init__()
        self.knn_context = _Holder()
        self.knn_context.distance_encoder = _Holder()
        self.knn_context.distance_encoder.mlp = nn.Sequential(
            *_conv_bn(10, 64, 0.01), *_conv_bn(64, 64, 0.01), nn.Conv2d(64, 128, kernel_size=1))
        self.knn_context.feat_conv = _EdgeConvParams(3, 128, 16)
        self.weight_unit = _Holder()
        self.weight_unit.mlp = nn.Sequential(
            *_conv_bn(256, 128, 0.01), *_conv_bn(128, 64, 0.01), nn.Conv2d(64, 32, kernel_size=1))


class CondList(list):
    """`cs` as returned by feat_extract: the 6 conditioning tensors plus the per-point conditioner
    outputs (cp, st) the flow kernels consume."""
    cp: Tensor
    st: Tensor


# ----------------------------------------------------------------------------------------
# HIP engine
# ----------------------------------------------------------------------------------------
def _host_state_dict(sd):
    """The state dict on the host with ONE device -> host copy per dtype (the weight folding reads ~400 small tensors: taken
    one at a time from the GPU that is ~400 synchronising copies, 0.09 s of the 0.19 s a plan takes to build)."""
    out, groups = {}, {}
    for k, v in sd.items():
        v = v.detach()
        if v.is_cuda:
            groups.setdefault(v.dtype, []).append((k, v))
        else:
            out[k] = v
    for dt, items in groups.items():
        flat = torch.cat([v.reshape(-1) for _, v in items]).cpu()
        pos = 0
        for k, v in items:
            out[k] = flat[pos:pos + v.numel()].reshape(v.shape)
            pos += v.numel()
    return {k: out[k] for k in sd}


class _Engine:
    def __init__(self, sd, device: torch.device, ec_mode: Optional[str] = None):
        sd = _host_state_dict(sd)
        self.lib = _lib.load()
        self.device = device
        # EdgeConv arithmetic (PointInterpFlow.ec_mode), both within the same 1e-5 parity bar:
        #   "f16n" (default, the product)  split-fp16 with a natural-scale low half, one accumulator, conv_out with swapped
        #                      operands (edgeconv4_kernel / edgeconv1n_kernel); needs |4^t x_t| < 65504 for growth layer t
        #                      (|activation| < 1023); out-of-range activations give inf/NaN, never a silently wrong number
        #   "f32"              v_mfma_f32_16x16x4_f32, bit-exact fp32 fma chain: the in-library A/B reference of the tests
        self.ec_mode = ec_mode or "f16n"
        if self.ec_mode not in ("f16n", "f32"):
            raise ValueError(f"unknown EdgeConv arithmetic mode {self.ec_mode!r}")
        pk = pack_plan(fold_state_dict(sd), self.ec_mode)
        self.blob = torch.from_numpy(pk["blob"]).to(device)
        self.base = self.blob.data_ptr()
        self.ec_tab0 = pk["ec_tab0"]
        self.ec_w = pk["ec_w"]
        self.ec4_w = pk["ec4_w"]
        self.ec1n_w = pk["ec1n_w"]
        self.post = [_lib.offsets(o) for o in pk["post"]]
        self.post_all = _lib.offsets([v for o in pk["post"] for v in o])
        self.flow = pk["flow"]
        self.interp_off = _lib.offsets(pk["interp"])
        self.ld_const = pk["ld_const"]
        self._logp_ws = {}
        self._side = None
        # interpolation weights as a parallel branch (side stream) + the weighted latent sum inside flow g: 1 = on, 0 = off.
        # Off by default: measured on MI355X it does not pay at any batch size (4 x 2048: 0.245 -> 0.255 ms per captured step;
        # 32 x 2048: unchanged) - the 152 KiB-LDS interpolation workgroups cannot share a CU with the EdgeConv chain's, so the
        # branch only time-slices the CUs and pays a fork / join.  Kept because a caller that has no other use for the wait
        # (e.g. z produced elsewhere, later) can start the weights early; bit-identical either way (tests/test_gpu_parity.py).
        self.split_interp = 0
        self.fuse_pq = -1          # pf_edgeconv_pq: -1 = fuse the P|Q GEMM into the EdgeConv launch for small batches, 0 never, 1 always

    def _p(self, off: int) -> int:
        return self.base + 4 * off

    def _edgeconv(self, u: int, src: int, idx16: Tensor, h: Tensor, B: int, N: int, s) -> None:
        """Fused EdgeConv unit u -> h [T, odim] in the engine's arithmetic (pf_edgeconv cfg table: include/puflow_hip.h)."""
        lib = self.lib
        if self.ec_mode == "f16n":
            w = self.ec1n_w[u] if u < 2 else self.ec4_w[u]
            rc = lib.pf_edgeconv(8 + u if u < 2 else 7, src, None, idx16.data_ptr(), self._p(w), h.data_ptr(), B, N, s)
        else:
            tab = self._p(self.ec_tab0) if u == 0 else None
            rc = lib.pf_edgeconv(_EC_CFG[u], src, tab, idx16.data_ptr(), self._p(self.ec_w[u]), h.data_ptr(), B, N, s)
        _lib.check(rc, f"pf_edgeconv[{self.ec_mode}][{u}]")

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def knn(self, xyz: Tensor) -> Tensor:
        B, N, _ = xyz.shape
        idx = torch.empty((B, N, 16), dtype=torch.int32, device=xyz.device)
        _lib.check(self.lib.pf_knn(xyz.data_ptr(), xyz.data_ptr(), B, N, N, 16, idx.data_ptr(), None, self._stream()),
                   "pf_knn")
        return idx

    def features(self, xyz: Tensor, idx16: Tensor, want_cs: bool, cs_only: bool = False):
        """6x (EdgeConv -> next unit's P|Q GEMM), then the six conditioner stages (they only feed the flow kernels, so they
        stay off the EdgeConv -> P|Q -> EdgeConv chain).  Returns cs (or None), cp [6,T,64], st [6,T,8].
        cs_only (with want_cs; the continuous model): the conditioner stage stops at the features - cp and st are None."""
        B, N, _ = xyz.shape
        T, dev, s = B * N, xyz.device, self._stream()
        cp = torch.empty((NUM_BLOCKS, T, 64), dtype=torch.float32, device=dev) if not (cs_only and want_cs) else None
        st = torch.empty((NUM_BLOCKS, T, 8), dtype=torch.float32, device=dev) if not (cs_only and want_cs) else None
        pq = torch.empty((T, 512), dtype=torch.float32, device=dev)
        hs: List[Tensor] = []
        # units read their P|Q table while the next one's is written: two tables, alternating (the fused kernel reads and writes
        # tables in the same launch; the two-kernel path would get by with one)
        pq2 = torch.empty((T, 512), dtype=torch.float32, device=dev)
        for u in range(NUM_BLOCKS):
            h = torch.empty((T, FEAT_CHANNELS[u + 1]), dtype=torch.float32, device=dev)
            src = xyz.data_ptr() if u == 0 else (pq if u % 2 else pq2).data_ptr()
            nxt = pq2 if u % 2 else pq
            if u + 1 == NUM_BLOCKS:
                self._edgeconv(u, src, idx16, h, B, N, s)
            elif self.ec_mode == "f16n":
                # EdgeConv u + the next unit's P|Q vectors: one launch for small batches, two kernels otherwise (pf_edgeconv_pq)
                w = self.ec1n_w[u] if u < 2 else self.ec4_w[u]
                _lib.check(self.lib.pf_edgeconv_pq(u, src, idx16.data_ptr(), self._p(w), h.data_ptr(), self.base, self.post[u],
                                                   nxt.data_ptr(), B, N, self.fuse_pq, s), f"pf_edgeconv_pq[{u}]")
            else:
                self._edgeconv(u, src, idx16, h, B, N, s)
                _lib.check(self.lib.pf_pq_gemm(u, h.data_ptr(), self.base, self.post[u], nxt.data_ptr(), T, s), f"pf_pq_gemm[{u}]")
            hs.append(h)
        cs: List[Optional[Tensor]] = [torch.empty((B, N, COND_CHANNELS[u]), dtype=torch.float32, device=dev) if want_cs else None
                                      for u in range(NUM_BLOCKS)]
        self._cond_all(hs, cs if want_cs else None, st, cp, T, s)
        return (cs if want_cs else None), cp, st

    def _cond_all(self, hs, cs, st: Optional[Tensor], cp: Optional[Tensor], T: int, s) -> None:
        """The six conditioner stages in one launch (pf_cond_all)."""
        import ctypes
        PtrArr = ctypes.c_void_p * NUM_BLOCKS
        h_arr = PtrArr(*[h.data_ptr() for h in hs])
        c_arr = PtrArr(*[c.data_ptr() for c in cs]) if cs is not None else None
        _lib.check(self.lib.pf_cond_all(h_arr, self.base, self.post_all, c_arr, st.data_ptr() if st is not None else None,
                                        cp.data_ptr() if cp is not None else None, T, s), "pf_cond_all")

    def logp_ws(self, B: int, N: int, dev) -> Tensor:
        """A zeroed workspace of pf_flow_fwd_logp (wave-tile sums + the arrival counter the kernel leaves at zero)."""
        return torch.zeros((int(self.lib.pf_flow_fwd_logp_ws_floats(B, N)),), dtype=torch.float32, device=dev)

    def flow_f(self, xyz: Tensor, cp: Tensor, st: Tensor, ws: Optional[Tensor] = None):
        B, N, _ = xyz.shape
        T, dev, s = B * N, xyz.device, self._stream()
        z = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        ld_pt = torch.empty((T,), dtype=torch.float32, device=dev)
        ldj = torch.empty((B,), dtype=torch.float32, device=dev)
        lps = torch.empty((B,), dtype=torch.float32, device=dev)
        logp = torch.empty((), dtype=torch.float32, device=dev)
        # flow f and the log-likelihood in one launch
        if ws is None:                                      # eager calls: one per shape and stream (a captured graph brings its own)
            key = (B, N, s)
            ws = self._logp_ws.get(key)
            if ws is None:
                ws = self._logp_ws[key] = self.logp_ws(B, N, dev)
        _lib.check(self.lib.pf_flow_fwd_logp(xyz.data_ptr(), cp.data_ptr(), st.data_ptr(), self._p(self.flow), z.data_ptr(),
                                             ld_pt.data_ptr(), self.ld_const, B, N, ldj.data_ptr(), lps.data_ptr(), logp.data_ptr(),
                                             ws.data_ptr(), s), "pf_flow_fwd_logp")
        return z, ldj, logp

    def interp(self, xyz: Tensor, z: Tensor, idx16: Tensor, R: int) -> Tensor:
        B, N, _ = xyz.shape
        u = torch.empty((B, N * R, 3), dtype=torch.float32, device=xyz.device)
        _lib.check(self.lib.pf_interp(xyz.data_ptr(), z.data_ptr(), idx16.data_ptr(), self.base, self.interp_off,
                                      u.data_ptr(), B, N, R, self._stream()), "pf_interp")
        return u

    def interp_weights(self, xyz: Tensor, idx16: Tensor) -> Tensor:
        """Softmax weights of the interpolation module, aw [B*N, 8, 4] (pf_interp_weights): no latents needed."""
        B, N, _ = xyz.shape
        aw = torch.empty((B * N, 8, 4), dtype=torch.float32, device=xyz.device)
        _lib.check(self.lib.pf_interp_weights(xyz.data_ptr(), idx16.data_ptr(), self.base, self.interp_off, aw.data_ptr(), B, N,
                                              self._stream()), "pf_interp_weights")
        return aw

    def flow_g_interp(self, aw: Tensor, z: Tensor, idx16: Tensor, cp: Tensor, st: Tensor, R: int) -> Tensor:
        """x = g(u) with u = the weighted latent sum formed inside the flow kernel (pf_flow_inv_interp, R <= 4)."""
        B, N, _ = z.shape
        x = torch.empty((B, N * R, 3), dtype=torch.float32, device=z.device)
        _lib.check(self.lib.pf_flow_inv_interp(aw.data_ptr(), z.data_ptr(), idx16.data_ptr(), cp.data_ptr(), st.data_ptr(),
                                               self._p(self.flow), x.data_ptr(), B, N, R, self._stream()), "pf_flow_inv_interp")
        return x

    def side_stream(self) -> "torch.cuda.Stream":
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def split_interp_for(self, T: int, R: int) -> bool:
        return R <= 4 and self.split_interp == 1

    def flow_g(self, u: Tensor, cp: Tensor, st: Tensor, R: int) -> Tensor:
        B, NR, _ = u.shape
        T = B * NR // R
        x = torch.empty_like(u)
        _lib.check(self.lib.pf_flow_inv(u.data_ptr(), cp.data_ptr(), st.data_ptr(), self._p(self.flow), x.data_ptr(),
                                        T, R, self._stream()), "pf_flow_inv")
        return x


    def profile_stages(self, xyz: Tensor, iters: int = 5, R: int = 4) -> dict:
        """Average ms per launch of each stage, timed with HIP events on the launch stream
        (torch's current stream IS the stream the kernels are enqueued on)."""
        B, N, _ = xyz.shape
        T, dev, s = B * N, xyz.device, self._stream()
        acc: dict = {}

        def timed(name, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            acc.setdefault(name, []).append((a, b))
            return r

        for _ in range(iters + 1):
            idx16 = timed("knn", lambda: self.knn(xyz))
            cp = torch.empty((NUM_BLOCKS, T, 64), dtype=torch.float32, device=dev)
            st = torch.empty((NUM_BLOCKS, T, 8), dtype=torch.float32, device=dev)
            pq = torch.empty((T, 512), dtype=torch.float32, device=dev)
            hs = []
            pq2 = torch.empty((T, 512), dtype=torch.float32, device=dev)
            fused = self.ec_mode == "f16n" and (self.fuse_pq == 1 or (self.fuse_pq < 0 and T <= 16384))
            for u in range(NUM_BLOCKS):
                h = torch.empty((T, FEAT_CHANNELS[u + 1]), dtype=torch.float32, device=dev)
                src = xyz.data_ptr() if u == 0 else (pq if u % 2 else pq2).data_ptr()
                nxt = pq2 if u % 2 else pq
                if fused and u + 1 < NUM_BLOCKS:
                    w = self.ec1n_w[u] if u < 2 else self.ec4_w[u]
                    timed(f"edgeconv{u}+pq", lambda: _lib.check(self.lib.pf_edgeconv_pq(
                        u, src, idx16.data_ptr(), self._p(w), h.data_ptr(), self.base, self.post[u], nxt.data_ptr(), B, N, 1, s)))
                else:
                    timed(f"edgeconv{u}", lambda: self._edgeconv(u, src, idx16, h, B, N, s))
                    if u + 1 < NUM_BLOCKS:
                        timed(f"pq{u}", lambda: _lib.check(self.lib.pf_pq_gemm(u, h.data_ptr(), self.base, self.post[u], nxt.data_ptr(), T, s)))
                hs.append(h)
            timed("cond_all", lambda: self._cond_all(hs, None, st, cp, T, s))
            z, _, _ = timed("flow_f+logp", lambda: self.flow_f(xyz, cp, st))
            u_ = timed("interp", lambda: self.interp(xyz, z, idx16, R))
            timed("flow_g", lambda: self.flow_g(u_, cp, st, R))
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v[1:]) / (len(v) - 1) for k, v in acc.items()}


# ----------------------------------------------------------------------------------------
class PointInterpFlow(nn.Module):
    """Reference surface: modules/discrete/interpflow.py:262-350."""

    def __init__(self, pc_channel: int = 3):
        super().__init__()
        if pc_channel != 3:
            raise ValueError("the HIP path is built for 3-D points (pc_channel=3)")
        self.num_blocks = NUM_BLOCKS
        self.num_neighbors = 16
        self.interp = _InterpParams()
        self.feat_convs = nn.ModuleList(
            [_EdgeConvParams(FEAT_CHANNELS[i], FEAT_CHANNELS[i + 1], GROWTH[i]) for i in range(NUM_BLOCKS)])
        self.merge_convs = nn.ModuleList(
            [_MergeParams(FEAT_CHANNELS[i + 1], COND_CHANNELS[i]) for i in range(NUM_BLOCKS)])
        self.flow_blocks = nn.ModuleList(
            [_FlowBlockParams(pc_channel, 64, COND_CHANNELS[i], i % 2 == 0) for i in range(NUM_BLOCKS)])
        self._engine_cache: Optional[_Engine] = None
        self.sync_batchnorm = False              # train-mode forward: BatchNorm statistics over all ranks (an argument, not a global)
        self.ec_mode: Optional[str] = None       # EdgeConv arithmetic: None / "f16n" = the product, "f32" = the exact A/B reference kernel

    # ---- plan cache ---------------------------------------------------------------------
    # The packed plan is a function of the parameters / buffers.  It is dropped by load_state_dict(), .to() / _apply(); every
    # use compares the tensors' addresses and version counters (every in-place update, e.g. an optimizer step or a BatchNorm
    # running-stat update, bumps them) and re-packs only when something changed - so eval() / train() toggles around
    # validation batches cost nothing.
    def invalidate_plan(self) -> None:
        self._engine_cache = None
        self._sig_tensors = None
        self._plan_gen = getattr(self, "_plan_gen", 0) + 1

    def load_state_dict(self, *a, **kw):
        self.invalidate_plan()
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self.invalidate_plan()
        return super()._apply(fn, *a, **kw)

    def _signature(self):
        """What the packed eval plan was built from: the address and the version counter of every parameter / buffer (the
        tensor LIST is cached: _apply() / load_state_dict() drop it).  Compared on EVERY use of the plan - ~800 attribute
        reads, tens of microseconds - so in-place edits and `.data` rebinds are seen in eval mode too.  Writers that go around
        torch's version counters must bump them (torch._C._increment_version): the fused optimizer
        (optim.FusedClipAdam.step_flat) and a replayed training graph (train_graph.GraphedTrainStep.__call__) do,
        dist.broadcast_* write with in-place copies; a train-mode forward, whose fused kernels update the BatchNorm running
        statistics through raw pointers, counts itself in `_train_forwards`.  (A write through `.data` is invisible to any
        such check: use `with torch.no_grad(): p.copy_(...)`.)"""
        ts = getattr(self, "_sig_tensors", None)
        if ts is None:
            ts = self._sig_tensors = list(self.parameters()) + list(self.buffers())
        # addresses are read on every use too: a rebind (`p.data = new_tensor`) changes the address without bumping _version
        return (tuple([t.data_ptr() for t in ts]), tuple([t._version for t in ts]), getattr(self, "_train_forwards", 0))

    def _engine(self, upratio: int = 4) -> _Engine:
        """The packed plan (it does not depend on the upsampling ratio; the argument is kept for callers of round 1)."""
        e = self._engine_cache
        dev = self.flow_blocks[0].actnorm.logs.device
        # always compared (a ~400-tuple, cheap against a forward): in-place edits in eval mode, or writes through `.data`
        # that a caller followed with torch._C._increment_version, are seen at the next use
        if e is not None and self._signature() != e.signature:
            self.invalidate_plan()
            e = None
        if e is not None and (e.device != dev or (self.ec_mode is not None and e.ec_mode != self.ec_mode)):
            self.invalidate_plan()
            e = None
        if e is None:
            if dev.type != "cuda":
                raise _lib.PuflowHipError("PointInterpFlow runs on the GPU only: move the module with .to('cuda')")
            e = _Engine(self.state_dict(), dev, self.ec_mode)
            e.signature = self._signature()
            self._engine_cache = e
        return e

    def _check_mode(self):
        if self.training:
            raise RuntimeError("this entry point is eval-only; in train() mode call forward() (differentiable path)")
        for b in self.flow_blocks:
            if not b.actnorm.is_inited:
                raise RuntimeError("ActNorm not initialised: load a checkpoint and call set_to_initialized_state() "
                                   "(reference upsample.py:32-33)")

    @staticmethod
    def _prep(xyz: Tensor) -> Tensor:
        if not xyz.is_cuda:
            raise _lib.PuflowHipError("input must be a GPU tensor (no CPU fallback)")
        return xyz.detach().contiguous().float()

    # ---- reference surface ---------------------------------------------------------------
    def set_to_initialized_state(self) -> None:
        for b in self.flow_blocks:
            b.actnorm.is_inited = True

    @torch.no_grad()
    def feat_extract(self, xyz: Tensor, knn_idx: Tensor) -> CondList:
        self._check_mode()
        xyz = self._prep(xyz)
        e = self._engine(4)
        cs, cp, st = e.features(xyz, knn_idx.to(torch.int32).contiguous(), want_cs=True)
        out = CondList(cs)
        out.cp, out.st = cp, st
        return out

    @torch.no_grad()
    def f(self, xyz: Tensor, cs: CondList) -> Tuple[Tensor, Tensor]:
        self._check_mode()
        z, ldj, _ = self._engine(4).flow_f(self._prep(xyz), cs.cp, cs.st)
        return z, ldj

    @torch.no_grad()
    def log_prob(self, xyz: Tensor, cs: CondList) -> Tuple[Tensor, Tensor]:
        self._check_mode()
        z, _, logp = self._engine(4).flow_f(self._prep(xyz), cs.cp, cs.st)
        return z, logp

    @torch.no_grad()
    def g(self, z: Tensor, cs: CondList, upratio: int) -> Tensor:
        """z: [B,N,3,R] (reference layout) -> [B,N*R,3]."""
        self._check_mode()
        u = torch.flatten(self._prep(z).transpose(2, 3), 1, 2).contiguous()
        return self._engine(upratio).flow_g(u, cs.cp, cs.st, upratio)

    def forward(self, xyz: Tensor, upratio: int = 4) -> Tuple[Tensor, Tensor]:
        if self.training:                       # differentiable path: BN batch statistics, ActNorm init, autograd
            from .train_ops import forward_train
            if not xyz.is_cuda:
                raise _lib.PuflowHipError("input must be a GPU tensor (no CPU fallback)")
            self._train_forwards = getattr(self, "_train_forwards", 0) + 1     # running statistics change: the eval plan is stale
            return forward_train(self, xyz, upratio)
        with torch.no_grad():
            return self._forward_eval(xyz, upratio)

    def _forward_eval(self, xyz: Tensor, upratio: int, ws: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        self._check_mode()
        xyz = self._prep(xyz)
        e = self._engine(upratio)
        idx16 = e.knn(xyz)
        if e.split_interp_for(xyz.shape[0] * xyz.shape[1], upratio):
            # the interpolation weights depend on xyz and the neighbour lists only: a parallel branch (side stream; inside a
            # captured graph a parallel branch of it) beside EdgeConv chain -> conditioners -> flow f; the weighted latent
            # sum happens inside the flow-g kernel.  Same bits as the fused order below.
            main, side = torch.cuda.current_stream(), e.side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                aw = e.interp_weights(xyz, idx16)
            _, cp, st = e.features(xyz, idx16, want_cs=False)
            z, _, logp = e.flow_f(xyz, cp, st, ws)
            main.wait_stream(side)
            aw.record_stream(main)
            x = e.flow_g_interp(aw, z, idx16, cp, st, upratio)
        else:
            _, cp, st = e.features(xyz, idx16, want_cs=False)
            z, _, logp = e.flow_f(xyz, cp, st, ws)
            u = e.interp(xyz, z, idx16, upratio)
            x = e.flow_g(u, cp, st, upratio)
        if _CHECK_FINITE and not bool(torch.isfinite(x).all() & torch.isfinite(logp)):
            # the split-fp16 kernels overflow to inf/NaN when an activation or weight leaves the fp16 range (65504):
            # loud by construction; this opt-in check (it synchronises) turns it into an error with the remedy
            raise _lib.PuflowHipError("non-finite output: an activation left the fp16 range of the split-fp16 kernels "
                                      "(or the input holds NaN/inf); check the input normalisation (patch.py:168-178)")
        return x, logp

    @torch.no_grad()
    def forward_stages(self, xyz: Tensor, upratio: int = 4) -> dict:
        """Every intermediate of forward() (parity tests)."""
        self._check_mode()
        xyz = self._prep(xyz)
        e = self._engine(upratio)
        idx16 = e.knn(xyz)
        cs, cp, st = e.features(xyz, idx16, want_cs=True)
        z, ldj, logp = e.flow_f(xyz, cp, st)
        u = e.interp(xyz, z, idx16, upratio)
        x = e.flow_g(u, cp, st, upratio)
        B, N, _ = xyz.shape
        fz = u.view(B, N, upratio, 3).transpose(2, 3)
        return dict(idx16=idx16, cs=cs, cp=cp, st=st, z=z, ldj=ldj, logp=logp, fz=fz, x=x)

    def sample(self, sparse: Tensor, upratio: int = 4) -> Tensor:
        dense, _ = self(sparse, upratio)
        return dense

    @torch.no_grad()
    def graphed(self, B: int, N: int, upratio: int = 4):
        """The eval forward for a fixed [B, N, 3] shape captured in a hipGraph: ONE launch per call instead of 11 - 16, so the
        step no longer depends on host launch latency / jitter (8 ranks sharing one host).  Returns `run(xyz) -> (x, logp)`;
        the results live in static buffers that the next call overwrites (clone them to keep them) and are
        bit-identical to `forward` (same kernels, same order).  `run.input` is the graph's own input buffer: a producer
        that writes its batch there and calls `run(run.input)` saves the device-to-device copy in front of every replay
        (a blit launch plus the idle gap around it: ~13 us, 5 % of a 4-patch step).
        The capture bakes device pointers into the packed weight blob: the callable keeps that plan alive, and refuses to
        replay (PuflowHipError) once the module's weights changed or moved (load_state_dict, .to(), an optimizer step in
        between) - capture again then."""
        self._check_mode()
        dev = self.flow_blocks[0].actnorm.logs.device
        static_in = torch.zeros((B, N, 3), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up outside capture: library load, plan packing
            self._forward_eval(static_in, upratio)
            ws = self._engine(upratio).logp_ws(B, N, dev)  # this graph's own log-likelihood workspace (graphs may replay concurrently)
        torch.cuda.current_stream(dev).wait_stream(side)
        engine = self._engine(upratio)                     # pinned: the graph reads this engine's blob
        from .train_graph import drain_collective_watchdog
        drain_collective_watchdog(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # see train_graph._CAPTURE_MODE: other threads (an RCCL
                                                                              # watchdog polling its events) must not break the capture
            out_x, out_logp = self._forward_eval(static_in, upratio, ws)
        if self._engine_cache is not engine:
            raise _lib.PuflowHipError("the plan was re-packed during capture")

        def run(xyz: Tensor) -> Tuple[Tensor, Tensor]:
            if tuple(xyz.shape) != (B, N, 3):
                raise ValueError(f"graph captured for {(B, N, 3)}, got {tuple(xyz.shape)}")
            if self.training or self._engine(upratio) is not engine:
                raise _lib.PuflowHipError("the captured graph is stale: the module's weights changed, moved or the module is "
                                          "in train() mode - call graphed() again")
            if xyz.data_ptr() != static_in.data_ptr():      # `run.input` handed back: the producer wrote the batch in place
                static_in.copy_(xyz)
            graph.replay()
            return out_x, out_logp

        run.input = static_in                                # zero-copy hand-over: fill this [B, N, 3] buffer, then run(run.input)
        run.graph = graph                                    # keep the capture, its plan and its workspace alive with the callable
        run.engine = engine
        run.ws = ws
        return run
