"""state_dict -> inference plan: folded dense matrices, then MFMA-fragment-ordered blobs.

Folds (all exact algebra), done once on the host in float64 and
rounded to fp32 once:
  * BatchNorm2d(eval) into the preceding 1x1 conv          (interpflow.py:203-221,91-98,144-151)
  * EdgeConv edge feature [x_i, x_j, x_j-x_i] -> per-point linear maps
        W.[x_i; x_j; x_j-x_i] = (W1-W3) x_i + (W2+W3) x_j   (interpflow.py:229-232)
    so every conv of a dense block gets its edge-feature part from two per-point vectors
    P[i] (with the bias) and Q[j]; only the growth-feature part runs per edge.
  * DistanceEncoder's first conv on [x_i, x_j, x_i-x_j, |x_i-x_j|] -> per-point P/Q + one
    column for the norm                                     (interpflow.py:106-113)
  * ActNorm o inv1x1 -> one 3x3 affine;  W^-1 precomputed   (normalize.py:34, permutate.py:118,123)
  * coupling1's first layer split into the h1 columns and the c columns
        W0.[h1; c] = W0[:, :tdim] h1 + W0[:, tdim:] c        (interpflow.py:38-41, coupling.py:114-118)
    the c part is per ORIGINAL point and shared by f and the R replicas of g.
  * WeightEstimationUnit's last conv truncated to its first R rows (interpflow.py:180).

`fold_state_dict` returns plain fp32 numpy matrices (tests/emulate_plan.py consumes them to
prove the algebra on CPU); `pack_plan` re-orders them into the 16x16x4 MFMA A-operand fragment
order used by csrc/ and concatenates everything into ONE device blob with an offset table.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

NUM_BLOCKS = 6
FEAT_CHANNELS = [3, 32, 64, 128, 128, 128, 128]
GROWTH = [8, 16, 32, 32, 32, 32]
COND_CHANNELS = [32, 64, 128, 128, 128, 128]
HDIM = 64
BN_EPS = 1e-5


def _np(sd, k):
    return sd[k].detach().cpu().numpy().astype(np.float64)      # numpy's cast: a torch .double() per tensor was half of the fold


def _fold_bn(sd, conv: str, bn: str):
    W = _np(sd, conv + ".weight")
    W = W.reshape(W.shape[0], -1)
    b = _np(sd, conv + ".bias")
    s = _np(sd, bn + ".weight") / np.sqrt(_np(sd, bn + ".running_var") + BN_EPS)
    return W * s[:, None], s * (b - _np(sd, bn + ".running_mean")) + _np(sd, bn + ".bias")


def fold_edgeconv(sd, pfx: str, C: int, g: int, nconv: int, odim: int, gpad: int) -> Dict[str, np.ndarray]:
    """Dense block -> per-point matrices PA (with bias pb) / QB [S,C] and per-edge growth
    matrices G[t] ([gpad, gpad*t], t=1..nconv-1) + Gout [odim, gpad*nconv].
    Row layout of S = gpad*nconv + odim: conv0 | conv1 | ... | conv_out; rows/cols beyond g
    inside each gpad block are zero (unit 0 has g=8, padded to 16 for the 16-row MFMA tile)."""
    S = gpad * nconv + odim
    PA = np.zeros((S, C)); QB = np.zeros((S, C)); pb = np.zeros(S)
    G: List[np.ndarray] = []
    for t in range(nconv + 1):
        if t < nconv:
            W, b = _fold_bn(sd, f"{pfx}.convs.{t}.0", f"{pfx}.convs.{t}.1")
            r0, rows = gpad * t, g
        else:
            W = _np(sd, pfx + ".conv_out.weight"); W = W.reshape(W.shape[0], -1)
            b = _np(sd, pfx + ".conv_out.bias")
            r0, rows = gpad * nconv, odim
        assert W.shape[1] == 3 * C + g * t
        W1, W2, W3, Gt = W[:, :C], W[:, C:2 * C], W[:, 2 * C:3 * C], W[:, 3 * C:]
        PA[r0:r0 + rows] = W1 - W3
        QB[r0:r0 + rows] = W2 + W3
        pb[r0:r0 + rows] = b
        if t >= 1:
            rp = gpad if t < nconv else odim
            Gp = np.zeros((rp, gpad * t))
            for u in range(t):
                Gp[:rows, gpad * u:gpad * u + g] = Gt[:, g * u:g * (u + 1)]
            G.append(Gp)
    out = {"PA": PA, "QB": QB, "pb": pb}
    for t, Gp in enumerate(G):
        out[f"G{t + 1}"] = Gp                         # G1..G{nconv-1}, G{nconv} = conv_out growth part
    return {k: v.astype(np.float32) for k, v in out.items()}


def _lin_a1d(sd, pfx):
    return (_np(sd, pfx + ".layers.0.weight"), _np(sd, pfx + ".layers.2.weight"), _np(sd, pfx + ".layers.2.bias"),
            _np(sd, pfx + ".layers.4.weight"), _np(sd, pfx + ".layers.4.bias"))


def fold_state_dict(sd, upratio: int = 4) -> Dict[str, object]:
    """All folded fp32 matrices of the inference plan (dense, un-permuted)."""
    plan: Dict[str, object] = {"upratio": upratio}
    units = []
    for i in range(NUM_BLOCKS):
        C, g, odim = FEAT_CHANNELS[i], GROWTH[i], FEAT_CHANNELS[i + 1]
        units.append(fold_edgeconv(sd, f"feat_convs.{i}", C, g, odim // g, odim, max(g, 16)))
    plan["units"] = units
    merges, flows = [], []
    for i in range(NUM_BLOCKS):
        p = f"merge_convs.{i}"
        merges.append({"W1": _np(sd, p + ".conv1.weight").astype(np.float32),
                       "b1": _np(sd, p + ".conv1.bias").astype(np.float32),
                       "W2": _np(sd, p + ".conv2.weight").astype(np.float32)})
        pf = f"flow_blocks.{i}"
        tdim = 1 if i % 2 == 0 else 2
        logs = _np(sd, pf + ".actnorm.logs").reshape(3)
        ab = _np(sd, pf + ".actnorm.bias").reshape(3)
        W = _np(sd, pf + ".permutate1.permutater.W")
        # forward:  p' = W (p*exp(logs) + ab) = A p + a0
        A = W * np.exp(logs)[None, :]
        a0 = W @ ab
        # inverse:  p = (W^-1 u - ab) * exp(-logs) = Ai u + ai0
        Winv = np.linalg.inv(W)
        Ai = Winv * np.exp(-logs)[:, None]
        ai0 = -ab * np.exp(-logs)
        ld_const = float(np.sum(logs) + np.linalg.slogdet(W)[1])      # per point
        c1 = _lin_a1d(sd, pf + ".coupling1.bias_net")
        sn = _lin_a1d(sd, pf + ".coupling2.scale_net")
        bn = _lin_a1d(sd, pf + ".coupling2.bias_net")
        flows.append({
            "tdim": tdim, "A": A.astype(np.float32), "a0": a0.astype(np.float32),
            "Ai": Ai.astype(np.float32), "ai0": ai0.astype(np.float32), "ld_const": ld_const,
            "c1_W0h": c1[0][:, :tdim].astype(np.float32), "c1_W0c": c1[0][:, tdim:].astype(np.float32),
            "c1_W2": c1[1].astype(np.float32), "c1_b2": c1[2].astype(np.float32),
            "c1_W4": c1[3].astype(np.float32), "c1_b4": c1[4].astype(np.float32),
            "s_W0": sn[0].astype(np.float32), "s_W2": sn[1].astype(np.float32), "s_b2": sn[2].astype(np.float32),
            "s_W4": sn[3].astype(np.float32), "s_b4": sn[4].astype(np.float32),
            "t_W0": bn[0].astype(np.float32), "t_W2": bn[1].astype(np.float32), "t_b2": bn[2].astype(np.float32),
            "t_W4": bn[3].astype(np.float32), "t_b4": bn[4].astype(np.float32),
        })
    plan["merges"], plan["flows"] = merges, flows
    # ---- interpolation module
    ip: Dict[str, np.ndarray] = {}
    p = "interp.knn_context.distance_encoder.mlp"
    W0, b0 = _fold_bn(sd, p + ".0", p + ".1")            # [64,10]: x_i(3) x_j(3) x_i-x_j(3) |.|(1)
    ip["d_PA"] = (W0[:, 0:3] + W0[:, 6:9]).astype(np.float32)   # multiplies x_i
    ip["d_QB"] = (W0[:, 3:6] - W0[:, 6:9]).astype(np.float32)   # multiplies x_j
    ip["d_wn"] = W0[:, 9].astype(np.float32)                    # multiplies ||x_i - x_j||
    ip["d_b0"] = b0.astype(np.float32)
    W3, b3 = _fold_bn(sd, p + ".3", p + ".4")
    ip["d_W3"], ip["d_b3"] = W3.astype(np.float32), b3.astype(np.float32)
    W6 = _np(sd, p + ".6.weight")
    ip["d_W6"], ip["d_b6"] = W6.reshape(W6.shape[0], -1).astype(np.float32), _np(sd, p + ".6.bias").astype(np.float32)
    ip["ec"] = fold_edgeconv(sd, "interp.knn_context.feat_conv", 3, 16, 8, 128, 16)
    p = "interp.weight_unit.mlp"
    W0, b0 = _fold_bn(sd, p + ".0", p + ".1")
    ip["w_W0"], ip["w_b0"] = W0.astype(np.float32), b0.astype(np.float32)
    W3, b3 = _fold_bn(sd, p + ".3", p + ".4")
    ip["w_W3"], ip["w_b3"] = W3.astype(np.float32), b3.astype(np.float32)
    W6 = _np(sd, p + ".6.weight"); W6 = W6.reshape(W6.shape[0], -1)
    # all r_max = 32 rows; a forward uses the first `upratio` of them (interpflow.py:180) - the plan does not depend on it
    ip["w_W6"], ip["w_b6"] = W6.astype(np.float32), _np(sd, p + ".6.bias").astype(np.float32)
    # linear o linear folds around the 256-channel context (no nonlinearity between the producers' last conv and the
    # weight unit's first conv, interpflow.py:134,144):  W0.[d; e] with d = W6 d2 + b6 and e = Gout feat + (PA x_i + QB x_j + pb)
    W0d = ip["w_W0"].astype(np.float64)
    W0a, W0b = W0d[:, :128], W0d[:, 128:]
    ec = ip["ec"]
    nrow = 16 * 8                                            # conv_out rows of the interp EdgeConv start at 128
    ip["f_dW"] = (W0a @ ip["d_W6"].astype(np.float64)).astype(np.float32)              # [128, 64]  on d2
    ip["f_b0"] = (ip["w_b0"].astype(np.float64) + W0a @ ip["d_b6"].astype(np.float64)).astype(np.float32)
    ip["f_eW"] = (W0b @ ec["G8"].astype(np.float64)).astype(np.float32)                # [128, 128] on the growth features
    ftab = np.zeros((128, 8), np.float64)
    ftab[:, 0:3] = W0b @ ec["PA"][nrow:].astype(np.float64)
    ftab[:, 3:6] = W0b @ ec["QB"][nrow:].astype(np.float64)
    ftab[:, 6] = W0b @ ec["pb"][nrow:].astype(np.float64)
    ip["f_tab"] = ftab.astype(np.float32)
    plan["interp"] = ip
    return plan


# ---------------------------------------------------------------------------------------
# MFMA fragment order (v_mfma_f32_16x16x4_f32 A operand; csrc/pf_mfma.h)
# ---------------------------------------------------------------------------------------
def frag_pack(W: np.ndarray) -> np.ndarray:
    """[OUT, IN] fp32 -> fragment order [OB][CB][lane 64][4] (rows/cols zero-padded to 16).

    Lane l of a wave holds, for 16-row block `ob` and 16-channel block `cb`, the four values
    W[ob*16 + (l & 15)][cb*16 + 4*(l >> 4) + r], r = 0..3 - i.e. the A operands of the four
    16x16x4 MFMA steps that consume accumulator register r of a 16-channel feature block."""
    out, inn = W.shape
    OB, CB = (out + 15) // 16, (inn + 15) // 16
    Wp = np.zeros((OB * 16, CB * 16), dtype=np.float32)
    Wp[:out, :inn] = W
    f = Wp.reshape(OB, 16, CB, 4, 4)            # ob,row,cb,q,r
    return np.ascontiguousarray(f.transpose(0, 2, 3, 1, 4)).reshape(OB, CB, 64, 4)


F16_LO_SCALE = 2048.0


def frag_pack_f16x2(W: np.ndarray) -> np.ndarray:
    """[OUT, IN] fp32 -> split-fp16 A-operand fragments of v_mfma_f32_16x16x32_f16, as a float32-typed byte
    image [OB][CP][2 splits hi / lo'][64 lanes][8 fp16], lo' = fp16((W - hi) * 2^11) (csrc/pf_mfma.h
    'split-fp16 path'; the continuous model's weight records).  Lane l (row = l & 15, q = l >> 4), element j of block pair
    cp: channel 32 cp + 4q + j (j < 4) or 32 cp + 16 + 4q + (j - 4) (j >= 4)."""
    out, inn = W.shape
    OB, CP = (out + 15) // 16, (inn + 31) // 32
    Wp = np.zeros((OB * 16, CP * 32), dtype=np.float32)
    Wp[:out, :inn] = W
    if np.abs(Wp).max(initial=0.0) >= 65504.0:
        raise ValueError("weight magnitude exceeds the fp16 range of the f16x2 path; use ec_mode='f32'")
    hi = Wp.astype(np.float16)
    lo = ((Wp - hi.astype(np.float32)).astype(np.float32) * np.float32(F16_LO_SCALE)).astype(np.float16)
    lanes = np.arange(64)
    row, q = lanes & 15, lanes >> 4
    j = np.arange(8)
    ch = np.where(j[None, :] < 4, 4 * q[:, None] + j[None, :], 16 + 4 * q[:, None] + (j[None, :] - 4))     # [64,8]
    res = np.zeros((OB, CP, 2, 64, 8), dtype=np.uint16)
    for si, part in enumerate((hi, lo)):
        bits = part.view(np.uint16).reshape(OB, 16, CP, 32)
        # lane = 16 q + row holds k = 16 h + 4 q + jj at slot j = 4 h + jj: a pure axis permutation of [OB,row,CP,h,q,jj]
        res[:, :, si] = bits.reshape(OB, 16, CP, 2, 4, 4).transpose(0, 2, 4, 1, 3, 5).reshape(OB, CP, 64, 8)
    return res.reshape(-1).view(np.float32)


def frag_pack_f16n(W: np.ndarray) -> np.ndarray:
    """[OUT, IN] fp32 -> split-fp16 fragments with a NATURAL-scale low half, lo = fp16(W - hi) (subnormals kept), same
    image layout as frag_pack_f16x2 ([OB][CP][hi / lo][64 lanes][8 fp16]).  For csrc/edgeconv.hip edgeconv4_kernel, where
    all three product terms share one accumulator; the caller scales W by exact powers of two (ec4_scales) so that the
    2^-25 absolute floor of a subnormal lo sits far below fp32 rounding."""
    out, inn = W.shape
    OB, CP = (out + 15) // 16, (inn + 31) // 32
    Wp = np.zeros((OB * 16, CP * 32), dtype=np.float32)
    Wp[:out, :inn] = W
    if np.abs(Wp).max(initial=0.0) >= 65504.0:
        raise ValueError("scaled weight magnitude exceeds the fp16 range of the f16n path; use ec_mode='f32'")
    hi = Wp.astype(np.float16)
    lo = (Wp - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    lanes = np.arange(64)
    row, q = lanes & 15, lanes >> 4
    j = np.arange(8)
    ch = np.where(j[None, :] < 4, 4 * q[:, None] + j[None, :], 16 + 4 * q[:, None] + (j[None, :] - 4))     # [64,8]
    res = np.zeros((OB, CP, 2, 64, 8), dtype=np.uint16)
    for si, part in enumerate((hi, lo)):
        bits = part.view(np.uint16).reshape(OB, 16, CP, 32)
        # lane = 16 q + row holds k = 16 h + 4 q + jj at slot j = 4 h + jj: a pure axis permutation of [OB,row,CP,h,q,jj]
        res[:, :, si] = bits.reshape(OB, 16, CP, 2, 4, 4).transpose(0, 2, 4, 1, 3, 5).reshape(OB, CP, 64, 8)
    return res.reshape(-1).view(np.float32)


def f16n_scale(W: np.ndarray) -> float:
    """Power of two that brings max |W| into [2^13, 2^14): the natural-scale low halves of (almost) all weights are then
    normal fp16 numbers and the 2^-25 absolute floor of the rest is < 2^-38 of the largest weight."""
    m = float(np.abs(W).max(initial=0.0))
    if m == 0.0 or not np.isfinite(m):
        return 1.0
    return float(2.0 ** int(np.clip(13 - np.floor(np.log2(m)), -24, 40)))


def frag_pack_f16n_scaled(W: np.ndarray):
    """-> (f16n fragment image of W * 2^sw, 2^-sw as float32): the kernel multiplies its accumulators by the second."""
    sc = f16n_scale(W)
    return frag_pack_f16n((W.astype(np.float64) * sc).astype(np.float32)), np.float32(1.0 / sc)


EC4_STEP = 4.0        # activation scale ratio between consecutive growth layers of edgeconv4_kernel (csrc/edgeconv.hip)


def ec4_scales(nconv: int = 4, g: int = 32, odim: int = 128):
    """Power-of-two scale plan of edgeconv4_kernel for one 128-channel unit.
    Feature block u is stored as a_u x_u with a_u = 4^u, so layer t's accumulators hold 4^t v_t and conv_out's 4^nconv y.
    Returns (row_scale_P [S], row_scale_Q [S], a [nconv + 1]): multipliers of the rows of the P|Q table (the P rows of
    conv_out stay unscaled: P_out is added after the max-pool, the kernel multiplies the pooled value by 4^-nconv)."""
    a = EC4_STEP ** np.arange(nconv + 1)
    S = g * nconv + odim
    rp = np.ones(S); rq = np.ones(S)
    for t in range(nconv):
        rp[g * t:g * (t + 1)] = a[t]; rq[g * t:g * (t + 1)] = a[t]
    rq[g * nconv:] = a[nconv]
    return rp.astype(np.float32), rq.astype(np.float32), a.astype(np.float32)


def ec4_weights(u: Dict[str, np.ndarray], nconv: int = 4, g: int = 32) -> np.ndarray:
    """Growth matrices G1..G_nconv of a folded 128-channel unit in the f16n image: columns of feature block c of layer t
    scaled by a_t / a_c (exact), then split hi / natural lo."""
    _, _, a = ec4_scales(nconv, g, u[f"G{nconv}"].shape[0])
    parts = []
    for t in range(1, nconv + 1):
        Gt = u[f"G{t}"].astype(np.float64).copy()
        for c in range(t):
            Gt[:, g * c:g * (c + 1)] *= a[t] / a[c]
        parts.append(frag_pack_f16n(Gt.astype(np.float32)))
    return np.concatenate(parts)


def frag_unpack_f16x2(F: np.ndarray, out: int, inn: int) -> np.ndarray:
    """Inverse of frag_pack_f16x2 (tests): float64 hi + lo' / 2^11 of the [out, inn] matrix."""
    OB, CP = (out + 15) // 16, (inn + 31) // 32
    img = np.ascontiguousarray(F, dtype=np.float32).view(np.uint16).reshape(OB, CP, 2, 64, 8).view(np.float16).astype(np.float64)
    val = img[:, :, 0] + img[:, :, 1] / F16_LO_SCALE                                   # [OB,CP,64,8]
    lanes = np.arange(64)
    row, q = lanes & 15, lanes >> 4
    j = np.arange(8)
    ch = np.where(j[None, :] < 4, 4 * q[:, None] + j[None, :], 16 + 4 * q[:, None] + (j[None, :] - 4))
    W = np.zeros((OB, 16, CP, 32))
    W[:, row[:, None], :, ch] = val.transpose(2, 3, 0, 1)
    return W.reshape(OB * 16, CP * 32)[:out, :inn]


def frag_unpack_f16n(F: np.ndarray, out: int, inn: int) -> np.ndarray:
    """Inverse of frag_pack_f16n (tests): float64 hi + lo of the [out, inn] matrix (still carrying the packer's scale)."""
    OB, CP = (out + 15) // 16, (inn + 31) // 32
    img = np.ascontiguousarray(F, dtype=np.float32).view(np.uint16).reshape(OB, CP, 2, 64, 8).view(np.float16).astype(np.float64)
    val = img[:, :, 0] + img[:, :, 1]
    lanes = np.arange(64)
    row, q = lanes & 15, lanes >> 4
    j = np.arange(8)
    ch = np.where(j[None, :] < 4, 4 * q[:, None] + j[None, :], 16 + 4 * q[:, None] + (j[None, :] - 4))
    W = np.zeros((OB, 16, CP, 32))
    W[:, row[:, None], :, ch] = val.transpose(2, 3, 0, 1)
    return W.reshape(OB * 16, CP * 32)[:out, :inn]


def frag_unpack(F: np.ndarray, out: int, inn: int) -> np.ndarray:
    OB, CB = F.shape[0], F.shape[1]
    f = F.reshape(OB, CB, 4, 16, 4).transpose(0, 3, 1, 2, 4)
    return f.reshape(OB * 16, CB * 16)[:out, :inn]


# ---------------------------------------------------------------------------------------
# Device blob: every folded matrix in kernel order + offset tables (float units, 64-aligned)
# ---------------------------------------------------------------------------------------
POST_SLOTS = ["M1", "b1", "M2", "H1", "S2", "bS2", "T2", "bT2", "ST4", "bST4", "PQ", "bPQ", "scales"]
POST_SCALES = ["M1", "M2", "H1", "S2", "T2", "ST4", "PQ"]      # order of the 2^-sw factors in the "scales" slot
INTERP_SLOTS = ["dtab", "d_W3", "d_b3", "d_W6", "scales", "ectab", "ec_w", "w_W0", "w_b0", "w_W3", "w_b3", "w_W6", "w_b6",
                "w_W6full", "w_b6full"]
INTERP_SCALES = ["dtab", "d_W3", "w1", "ec", "w_W3", "w_W6"]     # 2^-sw factors in the "scales" slot (csrc/interp.hip)
FLOW_REC = 5360


class _Blob:
    def __init__(self):
        self.parts: List[np.ndarray] = []
        self.n = 0

    def add(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        off = self.n
        pad = (-a.size) % 64
        self.parts.append(a)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.n += a.size + pad
        return off

    def data(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(0, np.float32)


def _pad_vec(v: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros(n, np.float32)
    out[:v.size] = v
    return out


def _edge_table(u: Dict[str, np.ndarray]) -> np.ndarray:
    """[S][8] rows: PA(3) QB(3) pb 0   (C = 3 units; csrc/edgeconv.hip C3 variant, interp.hip)."""
    S = u["PA"].shape[0]
    t = np.zeros((S, 8), np.float32)
    t[:, 0:3], t[:, 3:6], t[:, 6] = u["PA"], u["QB"], u["pb"]
    return t


def _etab_dense(xi: np.ndarray, xj: np.ndarray, nrm, const: np.ndarray) -> np.ndarray:
    """'Edge table' as a dense [rows, 32] matrix (column map of _etab_frag)."""
    rows = xi.shape[0]
    W = np.zeros((rows, 32), np.float32)
    W[:, 0:3] = xi
    W[:, 3] = xj[:, 0]
    W[:, 16:18] = xj[:, 1:3]
    if nrm is not None:
        W[:, 18] = nrm
    W[:, 19] = const
    return W


def _etab_frag(xi: np.ndarray, xj: np.ndarray, nrm, const: np.ndarray) -> np.ndarray:
    """'Edge table' [rows x e] as an f16x2 fragment image: the affine map of the raw edge inputs
    e = (x_i[3], x_j[3], |x_i - x_j|, 1).  The kernel (csrc/interp.hip) puts e into k-slots 0..7 of one 32-channel
    MFMA step = channels 0..3 of the pair's first block and 0..3 of its second block (columns 0..3 and 16..19)."""
    rows = xi.shape[0]
    W = np.zeros((rows, 32), np.float32)
    W[:, 0:3] = xi
    W[:, 3] = xj[:, 0]
    W[:, 16:18] = xj[:, 1:3]
    if nrm is not None:
        W[:, 18] = nrm
    W[:, 19] = const
    return frag_pack_f16x2(W)


def _ec_frags(u: Dict[str, np.ndarray], nconv: int) -> np.ndarray:
    return np.concatenate([frag_pack(u[f"G{t}"]).reshape(-1) for t in range(1, nconv + 1)])


def pack_flow_record(f: Dict[str, object]) -> np.ndarray:
    """One 5360-float flow-block record (layout: csrc/flow.hip header)."""
    rec = np.zeros(FLOW_REC, np.float32)
    img2, inv2 = frag_pack_f16n_scaled(f["c1_W2"])     # 4 ob x 2 pairs
    rec[0:4096] = img2
    W4 = f["c1_W4"]                                   # [3-td, 64]
    W4r = np.zeros((16, 64), np.float32)
    b4r = np.zeros(16, np.float32)
    for q in range(4):
        W4r[4 * q:4 * q + W4.shape[0]] = W4
        b4r[4 * q:4 * q + W4.shape[0]] = f["c1_b4"]
    img4, inv4 = frag_pack_f16n_scaled(W4r)           # 1 ob x 2 pairs
    rec[4096:5120] = img4
    rec[5120:5184] = f["c1_b2"] / inv2
    rec[5184:5200] = b4r / inv4
    W0h = np.zeros((64, 2), np.float32)
    W0h[:, :f["tdim"]] = f["c1_W0h"]
    rec[5200:5328] = W0h.reshape(-1)
    rec[5328:5337] = f["A"].reshape(-1); rec[5337:5340] = f["a0"]
    rec[5340:5349] = f["Ai"].reshape(-1); rec[5349:5352] = f["ai0"]
    rec[5352], rec[5353] = inv2, inv4
    return rec


def pack_plan(plan: Dict[str, object], ec_mode: str = "f16n") -> Dict[str, object]:
    """-> {'blob': fp32 1-D array, 'ec_tab0', 'ec_w'[6], 'post'[6][12], 'flow', 'interp'[13], 'ld_const'}.
    ec_mode "f16n" (the product): the P|Q rows are scaled for edgeconv4_kernel / edgeconv1n_kernel (ec4_scales);
    "f32" (the exact-fp32 A/B reference kernel) reads an unscaled table."""
    if ec_mode not in ("f16n", "f32"):
        raise ValueError(f"unknown EdgeConv arithmetic mode {ec_mode!r}")
    B = _Blob()
    out: Dict[str, object] = {}
    units = plan["units"]
    out["ec_tab0"] = B.add(_edge_table(units[0]))
    out["ec_w"] = [B.add(_ec_frags(units[i], 4)) for i in range(NUM_BLOCKS)]
    out["ec4_w"] = [None, None] + [B.add(ec4_weights(units[i])) for i in range(2, NUM_BLOCKS)]
    # f16n images of the narrow units 0 / 1 (csrc/edgeconv.hip edgeconv1n_kernel): G1 | G2 | G3 | Gout with the same
    # 4^t activation-scale plan; unit 0 carries its edge table with the rows scaled like the P|Q rows of the other units
    ec1n = []
    for i in range(2):
        parts = [ec4_weights(units[i], 4, 16)]
        if i == 0:
            rp, rq, _ = ec4_scales(4, 16, 32)
            # C = 3: every pre-activation (conv_out's P part included) comes from the table, so all its rows take the Q scale
            parts.append(frag_pack_f16n(_etab_dense(units[0]["PA"] * rq[:, None], units[0]["QB"] * rq[:, None], None, units[0]["pb"] * rq)))
        ec1n.append(B.add(np.concatenate(parts)))
    out["ec1n_w"] = ec1n + [None] * (NUM_BLOCKS - 2)
    out["ec_mode"] = ec_mode
    post = []
    for i in range(NUM_BLOCKS):
        m, f = plan["merges"][i], plan["flows"][i]
        # [192, cdim] first layers of the injector nets + coupling1's c-part, composed with the merge unit's bias-free
        # last Linear (c = W2 m, interpflow.py:258): linear o linear -> [192, odim/2] acting on m directly
        H1 = (np.concatenate([f["s_W0"], f["t_W0"], f["c1_W0c"]], axis=0).astype(np.float64)
              @ m["W2"].astype(np.float64)).astype(np.float32)
        ST4 = np.zeros((6, 128), np.float32)
        ST4[0:3, 0:64] = f["s_W4"]; ST4[3:6, 64:128] = f["t_W4"]
        bST4 = _pad_vec(np.concatenate([f["s_b4"], f["t_b4"]]), 16)
        # matrices: f16n fragment images (natural-scale low half) of W * 2^sw with sw per matrix; biases are stored
        # pre-multiplied by the same 2^sw (they initialise the accumulators) except bPQ, which is added after the
        # rescale; "scales" holds the 2^-sw factors in POST_SCALES order (csrc/pointwise.hip)
        inv = {}
        def img(name, W):
            im, inv[name] = frag_pack_f16n_scaled(W)
            return B.add(im)
        # M1 | H1 | S2 | T2 | ST4 back to back = the LDS image of cond_body (csrc/pointwise.hip): staged with ONE contiguous copy
        offs = {"M1": img("M1", m["W1"]), "H1": img("H1", H1), "S2": img("S2", f["s_W2"]), "T2": img("T2", f["t_W2"]),
                "ST4": img("ST4", ST4)}
        assert offs["ST4"] + 4 * 512 == B.n and (B.n - offs["M1"]) % 512 == 0
        offs["M2"] = img("M2", m["W2"])
        offs["b1"] = B.add(_pad_vec(m["b1"], ((m["b1"].size + 15) // 16) * 16) / inv["M1"])
        offs["bS2"] = B.add(f["s_b2"] / inv["S2"]); offs["bT2"] = B.add(f["t_b2"] / inv["T2"])
        offs["bST4"] = B.add(bST4 / inv["ST4"])
        if i + 1 < NUM_BLOCKS:
            nu = units[i + 1]
            Wpq = np.concatenate([nu["PA"], nu["QB"]], axis=0)
            bpq = np.concatenate([nu["pb"], np.zeros_like(nu["pb"])])
            if ec_mode == "f16n":                         # rows scaled by exact powers of two (edgeconv4 / edgeconv1n kernels)
                rp, rq, _ = ec4_scales() if i + 1 >= 2 else ec4_scales(4, 16, 64)
                rs = np.concatenate([rp, rq])
                Wpq, bpq = Wpq * rs[:, None], bpq * rs
            offs["PQ"] = img("PQ", Wpq)
            offs["bPQ"] = B.add(bpq)
        else:
            offs["PQ"], offs["bPQ"], inv["PQ"] = 0, 0, np.float32(1.0)
        offs["scales"] = B.add(_pad_vec(np.array([inv[k] for k in POST_SCALES], np.float32), 16))
        post.append([offs[k] for k in POST_SLOTS])
    out["post"] = post
    out["flow"] = B.add(np.concatenate([pack_flow_record(f) for f in plan["flows"]]))
    out["ld_const"] = float(sum(f["ld_const"] for f in plan["flows"]))
    ip = plan["interp"]
    W6r = np.zeros((16, 64), np.float32)           # R <= 4 fast path: rows 0..3 replicated into every 4-row q group
    b6r = np.zeros(16, np.float32)
    for q in range(4):
        W6r[4 * q:4 * q + 4] = ip["w_W6"][:4]
        b6r[4 * q:4 * q + 4] = ip["w_b6"][:4]
    ec = ip["ec"]
    ft = ip["f_tab"]
    # matrices: f16n fragment images (natural-scale low half) scaled by a power of two per GROUP of matrices that share
    # an accumulator; biases that initialise accumulators are pre-multiplied by the group's 2^sw; "scales" = the 2^-sw
    # factors in INTERP_SCALES order (csrc/interp.hip header lists the slots)
    def grp(*mats):
        sc = min(f16n_scale(m) for m in mats)
        return sc, [frag_pack_f16n((m.astype(np.float64) * sc).astype(np.float32)) for m in mats]
    dtab_d = _etab_dense(ip["d_PA"], ip["d_QB"], ip["d_wn"], ip["d_b0"])
    ectab_d = _etab_dense(ec["PA"][:128], ec["QB"][:128], None, ec["pb"][:128])
    w1tab_d = _etab_dense(ft[:, 0:3], ft[:, 3:6], None, ft[:, 6] + ip["f_b0"])     # W0b.(edge table) + b0 + W0a.b6
    s_dt, (i_dt,) = grp(dtab_d)
    s_d3, (i_d3,) = grp(ip["d_W3"])
    s_w1, (i_w1t, i_d6, i_w0) = grp(w1tab_d, ip["f_dW"], ip["f_eW"])
    ecG = [ec[f"G{t}"] for t in range(1, 8)]
    s_ec, ec_imgs = grp(ectab_d, *ecG)
    s_w3, (i_w3,) = grp(ip["w_W3"])
    s_w6, (i_w6r, i_w6f) = grp(W6r, ip["w_W6"])
    io = {}
    # the LDS image of interp_kernel, back to back in its order (4 + 8 + 8 + 8 + 16 + 16 + 16 = 76 fragment pairs of 2 KiB):
    # edge tables (distance encoder, growth pre-activations, w1 bracket), d_W3, the growth chain, (W0a W6), w_W3 - ONE copy
    for k, im in (("dtab", i_dt), ("ectab", ec_imgs[0]), ("w_b0", i_w1t), ("d_W3", i_d3), ("ec_w", np.concatenate(ec_imgs[1:])),
                  ("d_W6", i_d6), ("w_W3", i_w3)):
        io[k] = B.add(im)
    assert B.n - io["dtab"] == 76 * 512
    io.update({
        "d_b3": B.add(ip["d_b3"] * s_d3),
        "scales": B.add(_pad_vec(np.array([1 / s_dt, 1 / s_d3, 1 / s_w1, 1 / s_ec, 1 / s_w3, 1 / s_w6], np.float32), 16)),
        "w_W0": B.add(i_w0),                                                        # folded: W0b.Gout
        "w_b3": B.add(ip["w_b3"] * s_w3),
        "w_W6": B.add(i_w6r), "w_b6": B.add(b6r * s_w6),
        "w_W6full": B.add(i_w6f), "w_b6full": B.add(ip["w_b6"] * s_w6),
    })
    out["interp"] = [io[k] for k in INTERP_SLOTS]
    out["blob"] = B.data()
    return out


# ---------------------------------------------------------------------------------------
# Continuous (CNF) blocks: weight record of csrc/cnf.hip and the per-point context GEMM
# ---------------------------------------------------------------------------------------
CNF_REC = 10160
CNF_CTX = 288


def _np32(t) -> np.ndarray:
    return t.detach().cpu().numpy().astype(np.float32) if hasattr(t, "detach") else np.asarray(t, np.float32)


LOG2E = 1.4426950408889634


def pack_cnf_context(Hc: np.ndarray):
    """Hc [288, cd] of pack_cnf_block -> (f16n fragment image for pf_cnf_context, 2^-scale as float32)."""
    return frag_pack_f16n_scaled(np.ascontiguousarray(Hc, dtype=np.float32))


def cnf_split_ok(rec, T_end: float) -> bool:
    """May pf_cnf_steps take PF_CNF_SPLIT_GATES for this record?  The gate rows of its time vector already carry -log2e; the
    per-stage factor 2^(gt alpha h) has |h| <= T_end, and 2^100 x (any fp32 2^x) neither overflows to NaN nor loses a gate bit."""
    tv = np.asarray(rec[9872:9872 + CNF_CTX], np.float64)
    g = np.concatenate([tv[0:64], tv[128:192], tv[256:272]])
    return bool(np.abs(g).max() * abs(T_end) <= 100.0)


def pack_cnf_block(sd, i: int):
    """-> (rec [CNF_REC] fp32, Hc [288, cdim], hb [288], T_end).  A ConcatSquash layer (diffeq_layers.py:72-86) is
    (W x + b) * sigmoid(Wg [t; c] + bg) + Wb [t; c]: the c-columns of Wg / Wb form the context GEMM `Hc` (with bg
    as its bias), their t-columns the time coefficients inside the record (same 288-row layout as the GEMM output:
    gate1 bias1 gate2 bias2 gate3 bias3, the 3-row layer-3 pieces replicated into every 4-row q group)."""
    p = f"flow_blocks.{i}.cnf.odefunc.diffeq.layers"
    L = [{k: _np32(sd[f"{p}.{j}.{k}"]) for k in ("_layer.weight", "_layer.bias", "_hyper_bias.weight",
                                               "_hyper_gate.weight", "_hyper_gate.bias")} for j in range(3)]
    cdim = L[0]["_hyper_gate.weight"].shape[1] - 1
    Hc = np.zeros((CNF_CTX, cdim), np.float32)
    hb = np.zeros(CNF_CTX, np.float32)
    tv = np.zeros(CNF_CTX, np.float32)

    def put(row0, Wg, bg, rep):
        rows = Wg.shape[0]
        for qq in range(4 if rep else 1):
            r0 = row0 + 4 * qq if rep else row0
            Hc[r0:r0 + rows] = Wg[:, 1:]
            tv[r0:r0 + rows] = Wg[:, 0]
            if bg is not None:
                hb[r0:r0 + rows] = bg

    put(0, L[0]["_hyper_gate.weight"], L[0]["_hyper_gate.bias"], False)
    put(64, L[0]["_hyper_bias.weight"], None, False)
    put(128, L[1]["_hyper_gate.weight"], L[1]["_hyper_gate.bias"], False)
    put(192, L[1]["_hyper_bias.weight"], None, False)
    put(256, L[2]["_hyper_gate.weight"], L[2]["_hyper_gate.bias"], True)
    put(272, L[2]["_hyper_bias.weight"], None, True)
    # The kernel's sigmoid / tanh run on the hardware 2^x: sigmoid(a) = 1 / (1 + 2^(-log2e a)), tanh(a) = 1 - 2 / (2^(2 log2e a) + 1).
    # The constants are folded HERE (one multiply per gate and per tanh less on a VALU-bound kernel): gate rows carry
    # -log2e x their argument, the pre-activations of the two tanh layers (weights, biases, time / context bias rows) 2 log2e x.
    for lo, hi, f in ((0, 64, -LOG2E), (128, 192, -LOG2E), (256, 272, -LOG2E), (64, 128, 2 * LOG2E), (192, 256, 2 * LOG2E)):
        Hc[lo:hi] *= np.float32(f); hb[lo:hi] *= np.float32(f); tv[lo:hi] *= np.float32(f)

    rec = np.zeros(CNF_REC, np.float32)
    W1, W2, W3 = L[0]["_layer.weight"], L[1]["_layer.weight"], L[2]["_layer.weight"]
    rec[0:4096] = frag_pack_f16x2((W2.astype(np.float64) * (2 * LOG2E)).astype(np.float32))     # forward image: 2 log2e folded in
    rec[4096:8192] = frag_pack_f16x2(np.ascontiguousarray(W2.T))                                 # transposed image (VJP): plain
    W3r = np.zeros((16, 64), np.float32)
    b3r = np.zeros(16, np.float32)
    for qq in range(4):
        W3r[4 * qq:4 * qq + 3] = W3
        b3r[4 * qq:4 * qq + 3] = L[2]["_layer.bias"]
    rec[8192:9216] = frag_pack_f16x2(W3r)
    W1t = np.zeros((64, 4), np.float32); W1t[:, :3] = W1; W1t[:, 3] = L[0]["_layer.bias"]     # [W1 | b1]: the kernel's layer 1 is [W1 | b1] [y; 1]
    W1t = (W1t.astype(np.float64) * (2 * LOG2E)).astype(np.float32)                            # (the VJP divides its W1^T sums by 2 log2e again)
    W3t = np.zeros((64, 4), np.float32); W3t[:, :3] = W3.T
    rec[9216:9472] = W1t.reshape(-1)
    rec[9472:9728] = W3t.reshape(-1)
    rec[9728:9792] = L[0]["_layer.bias"]
    rec[9792:9856] = (L[1]["_layer.bias"].astype(np.float64) * (2 * LOG2E)).astype(np.float32)
    rec[9856:9872] = b3r
    rec[9872:10160] = tv
    T_end = float(_np32(sd[f"flow_blocks.{i}.cnf.sqrt_end_time"])) ** 2
    return rec, Hc, hb, T_end
