"""CPU oracle for the CONTINUOUS (CNF) PU-Flow x4 variant.  TEST INFRASTRUCTURE ONLY.

PARITY: the ODE right-hand side is PINNED, the ODE solver is UNPINNED.
  * `odenet` / `rhs` (ODEnet, ConcatSquashLinear, ODEfunc.forward incl. the Hutchinson divergence and the noise
    repetition of the inverse pass) are checked against golden vectors produced by the REFERENCE's own
    `modules/continuous/odefunc.py` + `diffeq_layers.py`, which import without torchdiffeq
    (tools/make_golden_cnf.py -> tests/golden/cnf_rhs.npz; tests/test_oracle_cnf.py, tests/test_gpu_cnf.py).
  * The solver cannot be pinned: the reference's `cnf.py:3-4` imports `torchdiffeq`, an un-vendored, un-pinned pip
    dependency (`docker/Dockerfile:43`) that is not installed and may not be stood in for, so `CNF`, `FlowBlock` and
    the whole continuous `PointInterpFlow.forward` cannot be run here.  For those, what follows is
  * a restatement of the reference's own modules (file:line cited per function), and
  * a restatement FROM THE PUBLISHED ALGORITHM of torchdiffeq's adaptive `dopri5` (0.2.x line:
    Dormand-Prince 5(4) tableau with FSAL, Hairer's initial step, RMS norm over the flattened state tuple,
    step controller safety 0.9 / ifactor 10 / dfactor 0.2, 4th-order dense output at the end time,
    reverse time by negating t and f),
anchored on the reference's call site (`cnf.py:97-113`: method dopri5, atol = rtol = 1e-5, states (x, logp, c)).
No golden vector of the reference exists for the integrated path; tests compare the HIP path against THIS file.

Shared with the discrete model (and pinned there): kNN, EdgeConv units, merge units, interpolation module
(`continuous/interpflow.py:14` imports them from the discrete file) - taken from oracle/ref_cpu.py.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ref_cpu as O

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NUM_BLOCKS = 6                      # continuous/interpflow.py:57
ATOL = RTOL = 1e-5                  # continuous/interpflow.py:28
LOG2PI = O.LOG2PI

# ---- Dormand-Prince 5(4) (torchdiffeq dopri5 tableau)
DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
DP_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]
DP_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]
SAFETY, IFACTOR, DFACTOR, ORDER = 0.9, 10.0, 0.2, 5


def interp_weights5(x: float, dt: float) -> Tuple[float, float, float, float, float]:
    """Dense output y(t0 + x dt) = w0 y0 + w1 y1 + wm y_mid + wf0 f0 + wf1 f1  (quartic fit `_interp_fit` /
    `_interp_evaluate`: a = 2dt(f1-f0) - 8(y1+y0) + 16ym, b = dt(5f0-3f1) + 18y0 + 14y1 - 32ym,
    c = dt(f1-4f0) - 11y0 - 5y1 + 16ym, d = dt f0, e = y0; y = (((a x + b) x + c) x + d) x + e)."""
    x2, x3, x4 = x * x, x * x * x, x * x * x * x
    return (-8 * x4 + 18 * x3 - 11 * x2 + 1, -8 * x4 + 14 * x3 - 5 * x2, 16 * x4 - 32 * x3 + 16 * x2,
            dt * (-2 * x4 + 5 * x3 - 4 * x2 + x), dt * (2 * x4 - 3 * x3 + x2))


class Dopri5Stats:
    def __init__(self):
        self.nfe = 0
        self.accepted = 0
        self.rejected = 0


def dopri5(func: Callable[[float, Tensor], Tensor], y0: Tensor, t0: float, t1: float, n_extra: int = 0,
           extra_d0_sumsq: float = 0.0, stats: Optional[Dopri5Stats] = None, rtol: float = RTOL, atol: float = ATOL) -> Tensor:
    """Adaptive Dormand-Prince from t0 to t1 > t0 on a state tensor y [rows, D].
    `n_extra` / `extra_d0_sumsq`: the reference integrates the tuple (x, logp, c) and torchdiffeq's RMS norm
    runs over the flattened tuple; the context c has zero derivative and zero error, so it only adds
    `n_extra` elements to every mean and sum((c / (atol + rtol |c|))^2) to the d0 term of the initial step."""
    st = stats or Dopri5Stats()
    n_tot = float(y0.numel() + n_extra)

    def rms(sumsq: float) -> float:
        return math.sqrt(sumsq / n_tot)

    f0 = func(t0, y0); st.nfe += 1
    # ---- initial step (Hairer, as in torchdiffeq `_select_initial_step`, order = 4)
    scale = atol + y0.abs() * rtol
    d0 = rms(float(((y0 / scale) ** 2).sum().double()) + extra_d0_sumsq)
    d1 = rms(float(((f0 / scale) ** 2).sum().double()))
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(t0 + h0, y0 + h0 * f0); st.nfe += 1
    d2 = rms(float((((f1 - f0) / scale) ** 2).sum().double())) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / ORDER)
    dt = min(100 * h0, h1)

    t = t0
    y = y0
    f = f0
    interp = None
    while t1 > t:                                                      # `_advance`: step until t1 is covered
        k = [f]
        for i in range(6):
            yi = y
            acc = None
            for j, bij in enumerate(DP_BETA[i]):
                if bij != 0:
                    acc = k[j] * bij if acc is None else acc + k[j] * bij
            yi = y + dt * acc
            k.append(func(t + DP_ALPHA[i] * dt, yi)); st.nfe += 1
        y1 = yi                                                         # FSAL: the last stage is the solution
        err = None
        for j, cj in enumerate(DP_C_ERR):
            if cj != 0:
                err = k[j] * cj if err is None else err + k[j] * cj
        err = dt * err
        tol = atol + rtol * torch.maximum(y.abs(), y1.abs())
        ratio = rms(float(((err / tol) ** 2).sum().double()))
        if not math.isfinite(ratio):                                    # torchdiffeq: NaN propagates / asserts
            raise FloatingPointError(f"dopri5: non-finite error norm at t = {t:g}")
        if not (t + dt > t):
            raise FloatingPointError(f"dopri5: underflow in dt ({dt:g}) at t = {t:g}")
        accept = ratio <= 1.0
        if accept:
            mid = None
            for j, cj in enumerate(DP_C_MID):
                if cj != 0:
                    mid = k[j] * cj if mid is None else mid + k[j] * cj
            interp = (y, y1, y + dt * mid, f, k[6], t, t + dt)
            t, y, f = t + dt, y1, k[6]
            st.accepted += 1
        else:
            st.rejected += 1
        if ratio == 0:
            dt = dt * IFACTOR
        else:
            dfac = 1.0 if ratio < 1 else DFACTOR
            dt = dt * min(IFACTOR, max(SAFETY / ratio ** (1.0 / ORDER), dfac))
    ya, yb, ym, fa, fb, ta, tb = interp
    w = interp_weights5((t1 - ta) / (tb - ta), tb - ta)
    return w[0] * ya + w[1] * yb + w[2] * ym + w[3] * fa + w[4] * fb


# --------------------------------------------------------------------------------------
# ODE right-hand side: ODEnet of ConcatSquashLinear layers, tanh (odefunc.py:60-104, diffeq_layers.py:72-86,
# continuous/interpflow.py:20-27: hdims (64, 64), layer_type concatsquash, nonlinearity tanh)
# --------------------------------------------------------------------------------------
def _csl(sd: SD, pfx: str, ctx: Tensor, x: Tensor) -> Tensor:
    gate = torch.sigmoid(F.linear(ctx, sd[pfx + "._hyper_gate.weight"], sd[pfx + "._hyper_gate.bias"]))
    bias = F.linear(ctx, sd[pfx + "._hyper_bias.weight"])
    return F.linear(x, sd[pfx + "._layer.weight"], sd[pfx + "._layer.bias"]) * gate + bias


def odenet(sd: SD, i: int, ctx: Tensor, y: Tensor) -> Tensor:
    p = f"flow_blocks.{i}.cnf.odefunc.diffeq.layers"
    h = torch.tanh(_csl(sd, p + ".0", ctx, y))
    h = torch.tanh(_csl(sd, p + ".1", ctx, h))
    return _csl(sd, p + ".2", ctx, h)


def rhs(sd: SD, i: int, t: float, state: Tensor, c: Tensor, e: Tensor) -> Tensor:
    """ODEfunc.forward (odefunc.py:121-148), conditional branch: state [rows, 4] = (y, logp) ->
    (dy, -e^T (d dy / d y) e).  The Hutchinson term is evaluated with autograd exactly like
    `divergence_approx` (odefunc.py:9-31)."""
    y = state[:, :3].detach().requires_grad_(True)
    tcol = torch.full((y.shape[0], 1), float(t), dtype=y.dtype)
    with torch.enable_grad():
        dy = odenet(sd, i, torch.cat([tcol, c], dim=-1), y)
        e_dzdx = torch.autograd.grad(dy, y, e)[0]
    div = (e_dzdx * e).sum(-1, keepdim=True)
    return torch.cat([dy.detach(), -div], dim=-1)


def rhs_vjp(sd: SD, i: int, t: float, y: Tensor, c: Tensor, e: Tensor, a_y: Tensor, a_l: Tensor):
    """Groundwork for the adjoint backward (torchdiffeq's odeint_adjoint @ cnf.py:89-99 evaluates exactly this product
    at every step of the backward integration): the vector-Jacobian product of the right-hand side (dy, -div) with the adjoint
    (a_y [rows,3], a_l [rows]) - the gradient of S = sum(a_y . dy - a_l e^T (d dy / d y) e) - WITHOUT autograd, by one
    forward pass of (value, tangent along e) and one reverse pass through the three ConcatSquash layers.
    Returns dict: y [rows,3], t (scalar), per layer l = 0..2: W, b ([out,in], [out]), gate_pre / bias_pre ([rows,out]: the
    gradients with respect to the hyper-networks' outputs, from which d/d(hyper weights) = pre^T [t, c] and d/dc follow).
    tests/test_oracle_cnf.py pins every entry to autograd."""
    p = f"flow_blocks.{i}.cnf.odefunc.diffeq.layers"
    rows = y.shape[0]
    ctx = torch.cat([torch.full((rows, 1), float(t), dtype=y.dtype), c], dim=-1)
    W = [sd[f"{p}.{l}._layer.weight"].to(y.dtype) for l in range(3)]
    b = [sd[f"{p}.{l}._layer.bias"].to(y.dtype) for l in range(3)]
    Wg = [sd[f"{p}.{l}._hyper_gate.weight"].to(y.dtype) for l in range(3)]
    bg = [sd[f"{p}.{l}._hyper_gate.bias"].to(y.dtype) for l in range(3)]
    Wb = [sd[f"{p}.{l}._hyper_bias.weight"].to(y.dtype) for l in range(3)]
    # forward: value x_l and tangent xd_l along e
    x, xd = [y], [e]
    lin, lind, g, pd = [], [], [], []
    for l in range(3):
        g.append(torch.sigmoid(F.linear(ctx, Wg[l], bg[l])))
        beta = F.linear(ctx, Wb[l])
        lin.append(F.linear(x[l], W[l], b[l]))
        lind.append(F.linear(xd[l], W[l]))
        pl, pdl = lin[l] * g[l] + beta, lind[l] * g[l]
        pd.append(pdl)
        if l < 2:
            xl = torch.tanh(pl)
            x.append(xl)
            xd.append((1 - xl * xl) * pdl)
        else:
            x.append(pl)
            xd.append(pdl)
    # reverse: seeds fbar = a_y, fdotbar = -a_l e
    xbar, xdbar = a_y, -a_l[:, None] * e
    out = {"t": y.new_zeros(())}
    for l in (2, 1, 0):
        if l < 2:
            xl = x[l + 1]
            pbar = (1 - xl * xl) * (xbar - 2 * xl * pd[l] * xdbar)
            pdbar = (1 - xl * xl) * xdbar
        else:
            pbar, pdbar = xbar, xdbar
        gl, gld = pbar * g[l], pdbar * g[l]
        out[f"W{l}"] = gl.t() @ x[l] + gld.t() @ xd[l]
        out[f"b{l}"] = gl.sum(0)
        gate_pre = (pbar * lin[l] + pdbar * lind[l]) * g[l] * (1 - g[l])
        out[f"gate_pre{l}"], out[f"bias_pre{l}"] = gate_pre, pbar
        out["t"] = out["t"] + (gate_pre * Wg[l][:, 0]).sum() + (pbar * Wb[l][:, 0]).sum()
        xbar, xdbar = gl @ W[l], gld @ W[l]
    out["y"] = xbar
    return out


def cnf_block(sd: SD, i: int, x: Tensor, c: Tensor, e: Tensor, reverse: bool, stats: Optional[Dopri5Stats] = None):
    """FlowBlock.forward / .inverse with batch_norm=False (continuous/interpflow.py:30-49) around CNF.forward
    (cnf.py:54-121).  x, e [rows, 3], c [rows, cdim] -> (x', delta_logp [rows])."""
    T = float(sd[f"flow_blocks.{i}.cnf.sqrt_end_time"]) ** 2           # cnf.py:75-78 (train_T)
    y0 = torch.cat([x, torch.zeros(x.shape[0], 1, dtype=x.dtype)], dim=-1)
    n_extra = c.numel()
    d0c = float(((c / (ATOL + RTOL * c.abs())) ** 2).sum().double())
    if not reverse:
        y1 = dopri5(lambda t, s: rhs(sd, i, t, s, c, e), y0, 0.0, T, n_extra, d0c, stats)
    else:
        # torchdiffeq integrates decreasing times as s = -t with f'(s, y) = -f(-s, y)
        y1 = dopri5(lambda s, st: -rhs(sd, i, -s, st, c, e), y0, -T, 0.0, n_extra, d0c, stats)
    return y1[:, :3], y1[:, 3]


@torch.no_grad()
def forward(sd: SD, xyz: Tensor, upratio: int = 4, noise: Optional[List[Tensor]] = None, stages: bool = False,
            dtype: torch.dtype = torch.float32):
    """continuous PointInterpFlow.forward (continuous/interpflow.py:116-126).  `noise[i]` [B,N,3] is block i's
    Hutchinson vector (the reference draws torch.randn_like lazily at the first RHS call of f and re-uses it,
    repeat_interleaved, in g: odefunc.py:134-137,11-12); default: drawn here from torch's global generator.
    `dtype=torch.float64` evaluates the same model (weights, inputs and noise are the fp32 values, exactly
    representable) with every operation of the network and the solver in double: the ANCHOR that tells the fp32
    oracle's own rounding error from a kernel's (tools/cnf_fp64_anchor.py).  The neighbour lists are always those of
    the fp32 distances - a discrete choice that is part of the model's input, not of its arithmetic."""
    xyz = xyz.float()
    B, N, _ = xyz.shape
    _, idx16 = O.knn_canonical(xyz, xyz, O.K_FEAT)
    idx8 = idx16[..., :O.K_INTERP].contiguous()
    if dtype != torch.float32:
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        xyz = xyz.to(dtype)
        noise = None if noise is None else [n.to(dtype) for n in noise]
    cs, _ = O.feat_extract(sd, xyz, idx16)
    if noise is None:
        noise = [torch.randn(B, N, 3).to(dtype) for _ in range(NUM_BLOCKS)]
    st = Dopri5Stats()
    # ---- f (continuous/interpflow.py:87-99)
    p = xyz.reshape(B * N, 3)
    ldj = torch.zeros(B, dtype=dtype)
    for i in range(NUM_BLOCKS):
        p, dl = cnf_block(sd, i, p, cs[i].reshape(B * N, -1), noise[i].reshape(B * N, 3), False, st)
        ldj = ldj + dl.view(B, N).sum(1)
    z = p.view(B, N, 3)
    logp = -torch.mean(torch.sum(-0.5 * (z ** 2 + LOG2PI), dim=(1, 2)) - ldj)        # interpflow.py:128-133, probs.py:87-93
    fz, a = O.interp(sd, z, xyz, idx8, upratio)
    # ---- g (continuous/interpflow.py:101-107)
    u = torch.flatten(fz.transpose(2, 3), 1, 2).reshape(B * N * upratio, 3)
    for i in reversed(range(NUM_BLOCKS)):
        c = torch.repeat_interleave(cs[i], upratio, dim=1).reshape(B * N * upratio, -1)
        e = torch.repeat_interleave(noise[i], upratio, dim=1).reshape(B * N * upratio, 3)
        u, _ = cnf_block(sd, i, u, c, e, True, st)
    x = u.view(B, N * upratio, 3)
    if stages:
        return dict(idx16=idx16, cs=cs, z=z, ldj=ldj, logp=logp, fz=fz, x=x, nfe=st.nfe, accepted=st.accepted,
                    rejected=st.rejected)
    return x, logp
