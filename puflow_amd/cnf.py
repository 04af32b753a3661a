"""Continuous (CNF) PU-Flow: `modules/continuous/interpflow.py` on the HIP library.

Same constructor, methods and the 390 state-dict keys of the reference's continuous `PointInterpFlow`
(`pretrain/puflow-x4-cnf-pu1k.pt` loads with `load_state_dict`).  Eval / inference only.

What runs where:
  * kNN, the six EdgeConv units, the merge units and the interpolation module are the discrete model's kernels
    (the reference imports those modules from the discrete file too, continuous/interpflow.py:14);
  * the per-point context terms of every ConcatSquash layer: one GEMM per block (`pf_gemm`);
  * every ODE right-hand side incl. the Hutchinson term, the Runge-Kutta stages (one fused launch per step attempt),
    the error / step norms: `pf_cnf_step`, `pf_cnf_rhs`, `pf_lincomb`, `pf_scaled_sumsq` (csrc/cnf.hip);
  * the adaptive step-size CONTROL of dopri5 (torchdiffeq semantics restated from its published algorithm, see
    oracle/cnf_ref.py header): a few host scalars per step - one device->host read of the error norm per step.

Parity: the right-hand side is pinned to the reference's own ODEfunc (tests/golden/cnf_rhs.npz); the solver is UNPINNED
(torchdiffeq is not installed and not vendored) - the integrated path is tested against oracle/cnf_ref.py, a from-text
restatement of dopri5.  Tolerance of the integrated path is the solver's own (atol = rtol = 1e-5).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .interpflow import (COND_CHANNELS, FEAT_CHANNELS, GROWTH, NUM_BLOCKS, _EdgeConvParams, _Engine, _InterpParams,
                         _MergeParams)
from .packing import CNF_CTX, cnf_split_ok, pack_cnf_block, pack_cnf_context
from .train_ops import _gemm
from .weights import state_dict_spec

ATOL = RTOL = 1e-5                       # continuous/interpflow.py:28
SAFETY, IFACTOR, DFACTOR, ORDER = 0.9, 10.0, 0.2, 5
MAX_NUM_STEPS = 100000                   # step attempts per integration; a bound, never reached by a sane model
DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]
DP_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


# ---- parameter holders (key names of the reference) ------------------------------------------------
class _ConcatSquashParams(nn.Module):
    """diffeq_layers.py:72-86."""

    def __init__(self, din: int, dout: int, dc: int):
        super().__init__()
        self._layer = nn.Linear(din, dout)
        self._hyper_bias = nn.Linear(1 + dc, dout, bias=False)
        self._hyper_gate = nn.Linear(1 + dc, dout)


class _ODEnetParams(nn.Module):
    def __init__(self, dc: int):
        super().__init__()
        self.layers = nn.ModuleList([_ConcatSquashParams(3, 64, dc), _ConcatSquashParams(64, 64, dc),
                                     _ConcatSquashParams(64, 3, dc)])


class _ODEfuncParams(nn.Module):
    def __init__(self, dc: int):
        super().__init__()
        self.diffeq = _ODEnetParams(dc)
        self.register_buffer("_num_evals", torch.tensor(0.))


class _CNFParams(nn.Module):
    def __init__(self, dc: int):
        super().__init__()
        self.register_parameter("sqrt_end_time", nn.Parameter(torch.sqrt(torch.tensor(0.5))))     # cnf.py:41
        self.odefunc = _ODEfuncParams(dc)


class _CNFBlockParams(nn.Module):
    def __init__(self, dc: int):
        super().__init__()
        self.cnf = _CNFParams(dc)


def _discrete_shell(sd) -> dict:
    """A discrete-model state dict around the shared extractor / interpolation weights: the discrete engine is reused
    for kNN, EdgeConv, merge (-> cs) and interpolation; its flow blocks get neutral parameters and are never run."""
    out = {}
    for key, shape, kind in state_dict_spec():
        if key in sd:
            out[key] = sd[key]
        elif kind == "inv1x1":
            out[key] = torch.eye(3)
        elif kind == "rev":
            out[key] = torch.tensor([2, 1, 0], dtype=torch.int64)
        elif kind == "nbt":
            out[key] = torch.tensor(0, dtype=torch.int64)
        else:
            out[key] = torch.zeros(shape)
    return out


class _CnfEngine:
    """Device-side plan of the continuous model + the dopri5 driver."""

    def __init__(self, sd, device: torch.device, upratio: int):
        self.lib = _lib.load()
        self.device = device
        self.R = upratio
        self.base = _Engine(_discrete_shell(sd), device)
        self.rec, self.Hc, self.Hci, self.hb, self.T_end, self.split = [], [], [], [], [], []
        for i in range(NUM_BLOCKS):
            rec, Hc, hb, T_end = pack_cnf_block(sd, i)
            self.rec.append(torch.from_numpy(rec).to(device))
            self.Hc.append(torch.from_numpy(Hc).to(device))
            img, inv = pack_cnf_context(Hc)
            self.Hci.append((torch.from_numpy(img).to(device), float(inv)))
            self.hb.append(torch.from_numpy(hb).to(device))
            self.T_end.append(T_end)
            # PF_CNF_SPLIT_GATES (include/puflow_hip.h): only where the factored 2^x cannot overflow; PF_CNF_SPLIT=0 keeps the plain kernel
            self.split.append(int(cnf_split_ok(rec, T_end) and os.environ.get("PF_CNF_SPLIT", "1") != "0"))
        self.ws = torch.empty(256, dtype=torch.float64, device=device)
        self.ws1k = torch.empty(1024, dtype=torch.float64, device=device)
        self.ws3k = torch.empty(3072, dtype=torch.float64, device=device)
        self.red = torch.empty(1, dtype=torch.float64, device=device)
        self.red3 = torch.empty(3, dtype=torch.float64, device=device)
        self.ctl = torch.zeros(16, dtype=torch.float64, device=device)       # dopri5 controller state (csrc/cnf.hip; zero before its first use)
        self.first_batch = int(os.environ.get("PF_CNF_FIRST_BATCH", "8"))    # step attempts enqueued before the first look
        self.next_batch = int(os.environ.get("PF_CNF_NEXT_BATCH", "4"))
        # attempts enqueued per integration when the whole forward runs WITHOUT reading the controller in between (attempts past
        # the end are ~4 us no-ops; an integration that needs more makes the forward fall back to the look-per-batch loop); 0 = off
        self.async_attempts = int(os.environ.get("PF_CNF_ASYNC_ATTEMPTS", "1"))             # 0 = never run blind
        self.logs = torch.zeros((2 * NUM_BLOCKS, 16), dtype=torch.float64, device=device)   # controller state after each integration
        # step attempts each of the twelve integrations took the last time (accepted + rejected): the blind run enqueues that
        # many + a margin (the trained model's blocks differ by 10x: 3 attempts for the short ones, 25 for the T = 36 block)
        self.hint: Optional[List[int]] = None
        self.nfe = 0
        self.accepted = 0
        self.rejected = 0

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    # ---- kernels ----------------------------------------------------------------------------------
    def context(self, i: int, c: Tensor) -> Tensor:
        """ctx [T,288] = c Hc^T + hb: everything a ConcatSquash layer takes from the context."""
        T, cd = c.shape
        ctx = torch.empty((T, CNF_CTX), dtype=torch.float32, device=c.device)
        if cd in (32, 64, 128) and c.is_contiguous():            # split-fp16 GEMM with register-resident weights (pf_pq_gemm's kernel)
            img, inv = self.Hci[i]
            _lib.check(self.lib.pf_cnf_context(c.data_ptr(), cd, img.data_ptr(), self.hb[i].data_ptr(), inv, ctx.data_ptr(), T,
                                               self._stream()), "pf_cnf_context")
        else:
            _gemm(c, cd, 1, self.Hc[i], 1, cd, ctx, CNF_CTX, self.hb[i], T, CNF_CTX, cd)
        return ctx

    def _rhs(self, i, y0, K, coef, h, t, sgn, ctx, e, kout, yout, rows, R):
        n = len(coef)
        arr = (ctypes.c_float * max(n, 1))(*[float(v) for v in coef])
        _lib.check(self.lib.pf_cnf_rhs(y0.data_ptr(), K.data_ptr() if n else None, arr, n, float(h), float(t), float(sgn),
                                       ctx.data_ptr(), e.data_ptr(), self.rec[i].data_ptr(), kout.data_ptr(),
                                       yout.data_ptr() if yout is not None else None, rows, R, self._stream()), "pf_cnf_rhs")
        self.nfe += 1

    def _lincomb(self, ptrs: List[Tensor], w: List[float], out: Tensor) -> None:
        n = len(ptrs)
        pa = (ctypes.c_void_p * n)(*[p.data_ptr() for p in ptrs])
        wa = (ctypes.c_float * n)(*[float(v) for v in w])
        _lib.check(self.lib.pf_lincomb(pa, wa, n, out.data_ptr(), out.numel(), self._stream()), "pf_lincomb")

    def _sumsq(self, a, b, s0, s1, K=None, w=None, h=0.0) -> float:
        n_terms = len(w) if w is not None else 0
        wa = (ctypes.c_float * max(n_terms, 1))(*([float(v) for v in w] if w is not None else [0.0]))
        _lib.check(self.lib.pf_scaled_sumsq(a.data_ptr() if a is not None else None, b.data_ptr() if b is not None else None,
                                            s0.data_ptr(), s1.data_ptr() if s1 is not None else None,
                                            K.data_ptr() if K is not None else None, wa, n_terms, float(h), RTOL, ATOL,
                                            s0.numel(), self.ws.data_ptr(), self.red.data_ptr(), self._stream()),
                   "pf_scaled_sumsq")
        return float(self.red.item())                       # the one device->host read per norm

    # ---- dopri5 (control flow of torchdiffeq's adaptive solver, restated: oracle/cnf_ref.py::dopri5) ------
    def context_norm(self, c: Tensor, out: Tensor) -> None:
        """out[0] (device double) = sum (c / (atol + rtol |c|))^2: the context's share of the solver's initial-step norm."""
        _lib.check(self.lib.pf_scaled_sumsq(c.data_ptr(), None, c.data_ptr(), None, None, (ctypes.c_float * 1)(0.0), 0, 0.0,
                                            RTOL, ATOL, c.numel(), self.ws.data_ptr(), out.data_ptr(), self._stream()),
                   "pf_scaled_sumsq")

    def integrate(self, i: int, x: Tensor, ctx: Tensor, e: Tensor, R: int, reverse: bool, extra_n: int,
                  extra_d0, log: Optional[Tensor] = None, blind: int = 0, extra_scale: float = 1.0) -> Tensor:
        """x [rows, >= 3] (row stride 3 or 4 floats: a previous block's state is taken as it lies) -> state [rows,4] =
        (x', delta logp) at the end time of block i.
        log (a [16] double device row, zero before its first use): do not look at the controller - `log` IS the controller
        state of this integration; enqueue `blind` attempts and return; the caller checks all rows once (`check_logs`).
        Otherwise the look-per-batch loop, which also leaves the number of attempts it took in `self.last_attempts`.
        extra_d0: a device double (or a python float / 0) x extra_scale - the context rows' share of |y0 / scale|^2."""
        rows = x.shape[0]
        dev = x.device
        T = self.T_end[i]
        t0, t1 = (0.0, T) if not reverse else (-T, 0.0)
        sgn = 1.0 if not reverse else -1.0
        n_tot = float(rows * 4 + extra_n)

        def net_t(s: float) -> float:                         # the time the network sees
            return s if not reverse else -s

        if x.dtype != torch.float32 or x.stride(1) != 1 or x.stride(0) not in (3, 4):
            x = x.float().contiguous()[:, :3].contiguous()
        bufs = torch.empty((5, rows, 4), dtype=torch.float32, device=dev)
        y, y1, f0, f1, out = bufs[0], bufs[1], bufs[2], bufs[3], bufs[4]
        # the state rows (x, 0), f0 and torchdiffeq's initial step size, on the device in two launches (csrc/cnf.hip:
        # pf_cnf_init); the controller state follows
        ctl = self.ctl if log is None else log
        if not isinstance(extra_d0, torch.Tensor):
            extra_d0 = torch.tensor([float(extra_d0)], dtype=torch.float64, device=dev) if extra_d0 else None
        _lib.check(self.lib.pf_cnf_init(ctl.data_ptr(), x.data_ptr(), int(x.stride(0)), y.data_ptr(), f0.data_ptr(),
                                        ctx.data_ptr(), e.data_ptr(), self.rec[i].data_ptr(), t0, t1, n_tot,
                                        extra_d0.data_ptr() if extra_d0 is not None else None, float(extra_scale),
                                        1 if reverse else 0, RTOL, ATOL, rows, R, self.ws3k.data_ptr(), self._stream()),
                   "pf_cnf_init")

        # ---- adaptive steps: ONE launch per attempt (six fused stage evaluations) + a one-wave controller kernel; the
        # accept / reject / next-dt decisions are taken on the device (csrc/cnf.hip: cnf_ctl_update, run by the workgroup of a step attempt that finishes last), the host enqueues a batch
        # of attempts and reads the controller state once per batch (attempts past the end of the integration are no-ops)
        if log is not None:
            _lib.check(self.lib.pf_cnf_steps(ctl.data_ptr(), y.data_ptr(), y1.data_ptr(), f0.data_ptr(), f1.data_ptr(),
                                             ctx.data_ptr(), e.data_ptr(), self.rec[i].data_ptr(), out.data_ptr(), RTOL, ATOL,
                                             rows, R, int(blind), self.ws1k.data_ptr(), self.split[i], self._stream()), "pf_cnf_steps")
            return out
        attempts = 0
        batch = self.first_batch
        while True:
            _lib.check(self.lib.pf_cnf_steps(ctl.data_ptr(), y.data_ptr(), y1.data_ptr(), f0.data_ptr(), f1.data_ptr(),
                                             ctx.data_ptr(), e.data_ptr(), self.rec[i].data_ptr(), out.data_ptr(), RTOL, ATOL,
                                             rows, R, batch, self.ws1k.data_ptr(), self.split[i], self._stream()), "pf_cnf_steps")
            attempts += batch
            st = ctl.cpu()                                      # the one device->host read per batch of attempts
            if st[5] != 0:
                break
            if attempts > MAX_NUM_STEPS:
                raise _lib.PuflowHipError(f"dopri5: more than {MAX_NUM_STEPS} step attempts in block {i}")
            batch = self.next_batch
        t, dt = float(st[0]), float(st[1])
        self.last_attempts = int(st[6]) + int(st[7])
        self.accepted += int(st[6])
        self.rejected += int(st[7])
        self.nfe += int(st[8])
        # torchdiffeq raises here too ('underflow in dt', NaN propagates into dt): without these guards a NaN error norm
        # makes every comparison False - no step is ever accepted and dt grows tenfold per attempt, forever
        if st[9] == 1:
            raise _lib.PuflowHipError(f"dopri5: non-finite error norm in block {i} at t = {t:g} (NaN / inf in the input, "
                                      "the weights or the state)")
        if st[9] == 2:
            raise _lib.PuflowHipError(f"dopri5: underflow in dt ({dt:g}) at t = {t:g}, block {i}")
        return out

    def time_step_attempts(self, i: int, x: Tensor, ctx: Tensor, e: Tensor, R: int, reverse: bool, extra_n: int, extra_d0,
                           attempts: int, extra_scale: float = 1.0):
        """bench.py's roofline leg: ONE integration of block i enqueued without a look at the controller, HIP events on the launch
        stream around the `attempts` step attempts alone (pf_cnf_steps: the kernel of the timed forward, cnf_step_dev_kernel) ->
        (milliseconds for all attempts, attempts the integration really took = accepted + rejected; the rest were no-ops)."""
        rows = x.shape[0]
        T = self.T_end[i]
        t0, t1 = (0.0, T) if not reverse else (-T, 0.0)
        x = x.float().contiguous()[:, :3].contiguous()
        bufs = torch.empty((5, rows, 4), dtype=torch.float32, device=x.device)
        y, y1, f0, f1, out = bufs[0], bufs[1], bufs[2], bufs[3], bufs[4]
        ctl = torch.zeros(16, dtype=torch.float64, device=x.device)
        _lib.check(self.lib.pf_cnf_init(ctl.data_ptr(), x.data_ptr(), int(x.stride(0)), y.data_ptr(), f0.data_ptr(),
                                        ctx.data_ptr(), e.data_ptr(), self.rec[i].data_ptr(), t0, t1, float(rows * 4 + extra_n),
                                        extra_d0.data_ptr() if extra_d0 is not None else None, float(extra_scale),
                                        1 if reverse else 0, RTOL, ATOL, rows, R, self.ws3k.data_ptr(), self._stream()), "pf_cnf_init")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(self.lib.pf_cnf_steps(ctl.data_ptr(), y.data_ptr(), y1.data_ptr(), f0.data_ptr(), f1.data_ptr(), ctx.data_ptr(),
                                         e.data_ptr(), self.rec[i].data_ptr(), out.data_ptr(), RTOL, ATOL, rows, R, int(attempts),
                                         self.ws1k.data_ptr(), self.split[i], self._stream()), "pf_cnf_steps")
        b.record()
        torch.cuda.synchronize()
        st = ctl.cpu()
        return a.elapsed_time(b), int(st[6]) + int(st[7]), bool(st[5] != 0)

    def check_logs(self, n: int, took: List[int]) -> int:
        """The controller states of the first n deferred integrations, ONE device -> host read: how many of them, from the
        first on, finished cleanly inside their attempts.  Their counters are added to the statistics and their attempt counts
        written into `took`."""
        L = self.logs[:n].cpu()
        ok = ((L[:, 5] != 0) & (L[:, 9] == 0)).tolist()
        k = 0
        while k < n and ok[k]:
            k += 1
        for j in range(k):
            took[j] = int(L[j, 6] + L[j, 7])
        self.accepted += int(L[:k, 6].sum())
        self.rejected += int(L[:k, 7].sum())
        self.nfe += int(L[:k, 8].sum())
        return k


class PointInterpFlow(nn.Module):
    """Reference surface: modules/continuous/interpflow.py:53-139 (the continuous `PointInterpFlow`)."""

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
        self.flow_blocks = nn.ModuleList([_CNFBlockParams(COND_CHANNELS[i]) for i in range(NUM_BLOCKS)])
        self._engine_cache: Optional[_CnfEngine] = None
        self.last_stats: dict = {}

    def invalidate_plan(self) -> None:
        self._engine_cache = None

    def load_state_dict(self, *a, **kw):
        self.invalidate_plan()
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self.invalidate_plan()
        return super()._apply(fn, *a, **kw)

    def _engine(self, upratio: int) -> _CnfEngine:
        e = self._engine_cache
        dev = self.flow_blocks[0].cnf.sqrt_end_time.device
        if e is None or e.R != upratio or e.device != dev:
            if dev.type != "cuda":
                raise _lib.PuflowHipError("PointInterpFlow runs on the GPU only: move the module with .to('cuda')")
            e = _CnfEngine({k: v.detach().cpu() for k, v in self.state_dict().items()}, dev, upratio)
            self._engine_cache = e
        return e

    def set_to_initialized_state(self) -> None:      # continuous/interpflow.py:113-114: nothing to initialise
        pass

    @torch.no_grad()
    def forward(self, xyz: Tensor, upratio: int = 4, noise: Optional[List[Tensor]] = None,
                stages: bool = False):
        """-> (x [B, N*upratio, 3], logp).  `noise[i]` [B,N,3]: block i's Hutchinson vector (default: torch.randn,
        like the reference, which draws it at the first right-hand-side call of f and re-uses it in g)."""
        if self.training:
            raise RuntimeError("the continuous model is built for inference only (call .eval())")
        if not xyz.is_cuda:
            raise _lib.PuflowHipError("input must be a GPU tensor (no CPU fallback)")
        xyz = xyz.detach().contiguous().float()
        B, N, _ = xyz.shape
        T = B * N
        eng = self._engine(upratio)
        eng.nfe = eng.accepted = eng.rejected = 0
        base = eng.base
        idx16 = base.knn(xyz)
        cs, _, _ = base.features(xyz, idx16, want_cs=True, cs_only=True)
        if noise is None:
            noise = [torch.randn(B, N, 3, device=xyz.device) for _ in range(NUM_BLOCKS)]
        es = [n.reshape(T, 3).contiguous().float() for n in noise]
        cflat = [c.reshape(T, -1).contiguous() for c in cs]
        ctx = [eng.context(i, cflat[i]) for i in range(NUM_BLOCKS)]
        # the context is a state of the reference's ODE (zero derivative): it only enters the solver's norms.  Six device words
        # (it was seven torch kernels per block and one host read in the middle of the forward)
        d0c = torch.empty(NUM_BLOCKS, dtype=torch.float64, device=xyz.device)
        for i in range(NUM_BLOCKS):
            eng.context_norm(cflat[i], d0c[i:i + 1])

        NI = 2 * NUM_BLOCKS

        def run(deferred: bool, k0: int = 0, outs=None):
            """f, interpolation, g = integrations 0 .. 11.  deferred: no look at a controller until all twelve are enqueued.
            k0 > 0: integrations < k0 are taken from `outs` (a blind run that got that far) and the rest is integrated."""
            outs = list(outs) if outs is not None else [None] * NI
            for i in range(NUM_BLOCKS):
                if i < k0:
                    continue
                p_in = xyz.reshape(T, 3) if i == 0 else outs[i - 1]        # [T,4]: a block reads the previous state's first three columns in place
                outs[i] = eng.integrate(i, p_in, ctx[i], es[i], 1, False, cflat[i].numel(), d0c[i:i + 1],
                                        eng.logs[i] if deferred else None, blind[i] if deferred else 0)
                took[i] = getattr(eng, "last_attempts", 0)
            # the six blocks' delta logp, summed per patch in one reduction (it was a reduce + an add per block)
            ldj = torch.stack([s_[:, 3] for s_ in outs[:NUM_BLOCKS]]).view(NUM_BLOCKS, B, N).sum(dim=(0, 2))
            z = outs[NUM_BLOCKS - 1][:, :3].reshape(B, N, 3)
            logp = -torch.mean(torch.sum(-0.5 * (z ** 2 + math.log(2 * math.pi)), dim=(1, 2)) - ldj)
            u0 = None
            for j, i in enumerate(reversed(range(NUM_BLOCKS))):
                k = NUM_BLOCKS + j
                if k < k0:
                    continue
                if j == 0:
                    u0 = base.interp(xyz, z.contiguous(), idx16, upratio).reshape(T * upratio, 3)        # row n*R + r
                outs[k] = eng.integrate(i, u0 if j == 0 else outs[k - 1], ctx[i], es[i], upratio, True, cflat[i].numel() * upratio,
                                        d0c[i:i + 1], eng.logs[k] if deferred else None, blind[k] if deferred else 0,
                                        extra_scale=float(upratio))
                took[k] = getattr(eng, "last_attempts", 0)
            return z, ldj, logp, outs[NI - 1][:, :3].contiguous(), outs

        # The host used to read the controller after every batch of attempts (~2 reads x 12 integrations, each a pipeline
        # bubble).  Now the whole forward is enqueued blind and the twelve final controller states are read ONCE.  An integration
        # that did not finish inside its attempts (or hit an error state) ends the blind part THERE: everything before it stands,
        # it and the integrations behind it go through the look-per-batch loop (which also raises the errors).  Same arithmetic
        # either way: attempts past the end of an integration are no-ops.
        # How many attempts to enqueue blind: what each integration took before + an eighth + 2; the counts decay by at most one
        # per forward, so an input stream whose counts flutter does not miss every other time (the first forward of an engine
        # takes the loop and leaves the counts).
        took: List[int] = [0] * NI
        k0, outs = 0, None
        ran_blind, enq = False, 0
        if eng.async_attempts > 0 and eng.hint is not None:
            blind = [h + h // 8 + 2 for h in eng.hint]
            ran_blind, enq = True, sum(blind)
            z, ldj, logp, u, outs = run(True)
            k0 = eng.check_logs(NI, took)
        if k0 < NI:
            blind = []
            z, ldj, logp, u, outs = run(False, k0, outs)
        old = eng.hint if eng.hint is not None else took
        eng.hint = [max(t, h - 1) for t, h in zip(took, old)]
        x = u.view(B, N * upratio, 3)
        # blind: was the forward enqueued without a look at a controller, how many step attempts were enqueued that way against the
        # attempts the twelve integrations took, and - when an integration did not finish inside its budget - from which
        # integration on the look-per-batch loop re-integrated (None: the blind forward stands)
        self.last_stats = dict(nfe=eng.nfe, accepted=eng.accepted, rejected=eng.rejected,
                               blind=dict(ran=ran_blind, attempts_enqueued=enq, attempts_taken=sum(took),
                                          fallback_from=(k0 if (ran_blind and k0 < NI) else None)))
        self.last_took = list(took)
        if stages:
            return dict(idx16=idx16, cs=cs, z=z, ldj=ldj, logp=logp, x=x, **self.last_stats)
        return x, logp

    def sample(self, sparse: Tensor, upratio: int = 4) -> Tensor:
        dense, _ = self(sparse, upratio)
        return dense
