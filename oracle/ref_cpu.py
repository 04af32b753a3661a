"""CPU oracle for the discrete PU-Flow x4 upsampling path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU *restatement* of the reference algorithm, written from the
equations in SURVEY.md Appendix A and checked against golden vectors captured from the
imported reference (tests/golden/*.npz, made by tools/make_golden.py).  It is the
checker for the HIP path.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it; nothing under puflow_amd/ does.

Parity status: PINNED for the network path (kNN -> EdgeConv -> flow f / log-prob ->
interpolation -> flow g) by golden vectors generated from the reference's own Python
(`/root/reference/modules/discrete/interpflow.py`) run in the build container.
The reference's third-party kernels (pytorch3d knn_points / chamfer_distance, kaolin
chamfer) are not vendored and not version-pinned by the reference (docker/Dockerfile:50,
`@stable`), so kNN tie order and Chamfer are *defined* here (see `knn_canonical`,
`chamfer_nn`) - "parity unpinned" for those two third-party boundaries only.

Everything works on a plain `state_dict` (the 408 reference keys); there is no nn.Module.
All functions cite the reference file:line they restate.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NUM_BLOCKS = 6           # interpflow.py:267
K_FEAT = 16              # interpflow.py:268
K_INTERP = 8             # interpflow.py:167 (constructor argument is overridden)
LOG2PI = float(math.log(2 * math.pi))   # probs.py:49


# --------------------------------------------------------------------------------------
# kNN  (pytorch3d.ops.knn_points call sites interpflow.py:104,:328)
# --------------------------------------------------------------------------------------
def pairwise_sqdist(p1: Tensor, p2: Tensor) -> Tensor:
    """Unfused fp32 squared distance ((dx*dx)+(dy*dy))+(dz*dz); [B,N,M].

    Each torch elementwise op rounds separately, so no FMA contraction happens here;
    the HIP kernel uses __fmul_rn/__fadd_rn to produce the same bits.
    """
    d = p1[:, :, None, :] - p2[:, None, :, :]
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return (dx * dx + dy * dy) + dz * dz


def knn_canonical(p1: Tensor, p2: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """Exact brute-force kNN ordered by (distance asc, index asc).  -> dists [B,N,K], idx int64."""
    d = pairwise_sqdist(p1.float(), p2.float())
    ds, order = torch.sort(d, dim=-1, stable=True)       # stable => index-ascending on ties
    return ds[..., :K].contiguous(), order[..., :K].contiguous()


def knn_gather(x: Tensor, idx: Tensor) -> Tensor:
    """out[b,n,k,:] = x[b, idx[b,n,k], :]   (interpflow.py:229)."""
    B = x.shape[0]
    return x[torch.arange(B).view(B, 1, 1), idx]


# --------------------------------------------------------------------------------------
# EdgeConv dense block  (FeatureExtractUnit, interpflow.py:190-248)
# --------------------------------------------------------------------------------------
def _conv_bn_lrelu(sd: SD, pfx: str, f: Tensor, slope: float, training: bool = False) -> Tensor:
    y = F.conv2d(f, sd[pfx + ".0.weight"], sd[pfx + ".0.bias"])
    y = F.batch_norm(y, sd[pfx + ".1.running_mean"], sd[pfx + ".1.running_var"],
                     sd[pfx + ".1.weight"], sd[pfx + ".1.bias"], training=False, eps=1e-5)
    return F.leaky_relu(y, slope)


def edgeconv_unit(sd: SD, pfx: str, x: Tensor, idx: Tensor, pooling: bool = True) -> Tensor:
    """x [B,N,C], idx [B,N,K] -> pooled [B,N,odim] or un-pooled [B,odim,N,K]."""
    nconv = 0
    while f"{pfx}.convs.{nconv}.0.weight" in sd:
        nconv += 1
    nb = knn_gather(x, idx)                                   # [B,N,K,C]  x_j
    xi = x.unsqueeze(2).expand_as(nb)                         # x_i
    f = torch.cat([xi, nb, nb - xi], dim=-1).permute(0, 3, 1, 2)   # [B,3C,N,K]  (:229-236)
    for t in range(nconv):
        g = _conv_bn_lrelu(sd, f"{pfx}.convs.{t}", f, 0.05)
        f = torch.cat([f, g], dim=1)                          # old first, growth last (:240)
    y = F.conv2d(f, sd[pfx + ".conv_out.weight"], sd[pfx + ".conv_out.bias"])
    if not pooling:
        return y
    return y.max(dim=-1)[0].transpose(1, 2)                   # [B,N,odim]  (:245-246)


def feat_merge(sd: SD, i: int, h: Tensor) -> Tensor:
    """FeatMergeUnit (interpflow.py:251-258)."""
    p = f"merge_convs.{i}"
    return F.linear(F.relu(F.linear(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])),
                    sd[p + ".conv2.weight"])


def feat_extract(sd: SD, xyz: Tensor, knn_idx: Tensor) -> Tuple[List[Tensor], List[Tensor]]:
    """PointInterpFlow.feat_extract (interpflow.py:292-300). Returns (cs, pooled hs)."""
    cs, hs = [], []
    h = xyz
    for i in range(NUM_BLOCKS):
        h = edgeconv_unit(sd, f"feat_convs.{i}", h, knn_idx)
        hs.append(h)
        cs.append(feat_merge(sd, i, h))
    return cs, hs


# --------------------------------------------------------------------------------------
# Flow blocks (interpflow.py:46-82, coupling.py, normalize.py, permutate.py)
# --------------------------------------------------------------------------------------
def lin_a1d(sd: SD, pfx: str, h: Tensor) -> Tensor:
    """LinearA1D (interpflow.py:22-43): Linear(no bias)->LReLU(.01)->Linear->LReLU->Linear."""
    h = F.leaky_relu(F.linear(h, sd[pfx + ".layers.0.weight"]), 0.01)
    h = F.leaky_relu(F.linear(h, sd[pfx + ".layers.2.weight"], sd[pfx + ".layers.2.bias"]), 0.01)
    return F.linear(h, sd[pfx + ".layers.4.weight"], sd[pfx + ".layers.4.bias"])


def _split(p: Tensor, i: int):
    t = 1 if i % 2 == 0 else 2                                 # coupling.py:114-118
    return p[..., :t], p[..., t:]


def flow_block_forward(sd: SD, i: int, p: Tensor, c: Tensor) -> Tuple[Tensor, Tensor]:
    """FlowBlock.forward (interpflow.py:66-74). p [B,N,3], c [B,N,cdim] -> (p', logdet [B])."""
    pf = f"flow_blocks.{i}"
    N = p.shape[1]
    logs, bias = sd[pf + ".actnorm.logs"], sd[pf + ".actnorm.bias"]
    p = p * torch.exp(logs) + bias                              # normalize.py:34
    ld = torch.sum(logs) * N
    W = sd[pf + ".permutate1.permutater.W"]
    p = torch.einsum("ij,bnj->bni", W, p)                       # permutate.py:118
    ld = ld + torch.slogdet(W)[1] * N
    h1, h2 = _split(p, i)
    h2 = h2 - lin_a1d(sd, pf + ".coupling1.bias_net", torch.cat([h1, c], dim=-1))
    p = torch.cat([h1, h2], dim=-1)
    p = p[:, :, sd[pf + ".permutate2.permutater.direct_idx"]]   # permutate.py:77
    s = lin_a1d(sd, pf + ".coupling2.scale_net", c)
    t = lin_a1d(sd, pf + ".coupling2.bias_net", c)
    p = (p - t) * torch.exp(-s)                                 # coupling.py:136
    ld = ld + (-torch.sum(torch.flatten(s, start_dim=1), dim=1))
    return p, ld


def flow_block_inverse(sd: SD, i: int, z: Tensor, c: Tensor) -> Tensor:
    """FlowBlock.inverse (interpflow.py:76-82)."""
    pf = f"flow_blocks.{i}"
    s = lin_a1d(sd, pf + ".coupling2.scale_net", c)
    t = lin_a1d(sd, pf + ".coupling2.bias_net", c)
    z = z * torch.exp(s) + t                                    # coupling.py:149
    z = z[:, :, sd[pf + ".permutate2.permutater.inverse_idx"]]
    h1, h2 = _split(z, i)
    h2 = h2 + lin_a1d(sd, pf + ".coupling1.bias_net", torch.cat([h1, c], dim=-1))
    z = torch.cat([h1, h2], dim=-1)
    W = sd[pf + ".permutate1.permutater.W"]
    z = torch.einsum("ij,bnj->bni", torch.inverse(W), z)        # permutate.py:123-124
    logs, bias = sd[pf + ".actnorm.logs"], sd[pf + ".actnorm.bias"]
    return (z - bias) * torch.exp(-logs)                        # normalize.py:41


def flow_f(sd: SD, xyz: Tensor, cs: List[Tensor]):
    """PointInterpFlow.f (interpflow.py:302-313) -> z, log_det_J [B], per-block (p_i, ld_i)."""
    B = xyz.shape[0]
    ldj = torch.zeros(B)
    p = xyz
    trace = []
    for i in range(NUM_BLOCKS):
        p, ld = flow_block_forward(sd, i, p, cs[i])
        ldj = ldj + ld
        trace.append((p, ld))
    return p, ldj, trace


def log_prob(sd: SD, xyz: Tensor, cs: List[Tensor]):
    """PointInterpFlow.log_prob (interpflow.py:339-345) + standard_logp (probs.py:87-93)."""
    z, ldj, _ = flow_f(sd, xyz, cs)
    lp = torch.sum(-0.5 * (z ** 2 + LOG2PI), dim=(1, 2))
    return z, -torch.mean(lp + ldj), ldj


def flow_g(sd: SD, fz: Tensor, cs: List[Tensor], upratio: int) -> Tensor:
    """PointInterpFlow.g (interpflow.py:315-321). fz [B,N,3,R] -> [B,N*R,3]."""
    z = torch.flatten(fz.transpose(2, 3), 1, 2)
    for i in reversed(range(NUM_BLOCKS)):
        c = torch.repeat_interleave(cs[i], upratio, dim=1)
        z = flow_block_inverse(sd, i, z, c)
    return z


# --------------------------------------------------------------------------------------
# Interpolation module (interpflow.py:85-186)
# --------------------------------------------------------------------------------------
def _conv_bn_lrelu01(sd: SD, pfx: str, a: int, f: Tensor) -> Tensor:
    y = F.conv2d(f, sd[f"{pfx}.{a}.weight"], sd[f"{pfx}.{a}.bias"])
    b = a + 1
    y = F.batch_norm(y, sd[f"{pfx}.{b}.running_mean"], sd[f"{pfx}.{b}.running_var"],
                     sd[f"{pfx}.{b}.weight"], sd[f"{pfx}.{b}.bias"], training=False, eps=1e-5)
    return F.leaky_relu(y, 0.01)


def interp_weights(sd: SD, xyz: Tensor, idx8: Tensor, upratio: int) -> Tensor:
    """Softmax interpolation weights [B,N,R,k] (interpflow.py:100-159,:180)."""
    nb = knn_gather(xyz, idx8)                                  # [B,N,k,3]
    xi = xyz.unsqueeze(2).expand_as(nb)
    vec = xi - nb                                               # x_i - x_j (:110)
    dist = torch.sqrt(torch.sum(vec ** 2, dim=-1, keepdim=True))
    fd = torch.cat([xi, nb, vec, dist], dim=-1).permute(0, 3, 1, 2)   # [B,10,N,k]
    p = "interp.knn_context.distance_encoder.mlp"
    d = _conv_bn_lrelu01(sd, p, 0, fd)
    d = _conv_bn_lrelu01(sd, p, 3, d)
    d = F.conv2d(d, sd[p + ".6.weight"], sd[p + ".6.bias"])     # [B,128,N,k]
    feat = edgeconv_unit(sd, "interp.knn_context.feat_conv", xyz, idx8, pooling=False)
    ctx = torch.cat([d, feat], dim=1)                           # [B,256,N,k]  (:134)
    p = "interp.weight_unit.mlp"
    w = _conv_bn_lrelu01(sd, p, 0, ctx)
    w = _conv_bn_lrelu01(sd, p, 3, w)
    w = F.conv2d(w, sd[p + ".6.weight"], sd[p + ".6.bias"])     # [B,32,N,k]
    w = w.permute(0, 2, 1, 3)                                   # [B,N,32,k]
    return F.softmax(w[:, :, :upratio], dim=-1)


def interp(sd: SD, z: Tensor, xyz: Tensor, idx8: Tensor, upratio: int):
    """InterpolationModule.forward (interpflow.py:173-186) -> fz [B,N,3,R], weights."""
    a = interp_weights(sd, xyz, idx8, upratio)
    nz = knn_gather(z, idx8).permute(0, 1, 3, 2)                # [B,N,3,k]
    return torch.einsum("bnck,bnrk->bncr", nz, a), a


# --------------------------------------------------------------------------------------
# Whole forward (interpflow.py:327-337)
# --------------------------------------------------------------------------------------
@torch.no_grad()
def forward(sd: SD, xyz: Tensor, upratio: int = 4, stages: bool = False):
    xyz = xyz.float()
    _, idx16 = knn_canonical(xyz, xyz, K_FEAT)
    idx8 = idx16[..., :K_INTERP].contiguous()   # first 8 of the (dist,idx)-sorted 16 == canonical kNN8
    cs, hs = feat_extract(sd, xyz, idx16)
    z, logp, ldj = log_prob(sd, xyz, cs)
    fz, a = interp(sd, z, xyz, idx8, upratio)
    x = flow_g(sd, fz, cs, upratio)
    if stages:
        return dict(idx16=idx16, idx8=idx8, cs=cs, hs=hs, z=z, logp=logp, ldj=ldj, w=a, fz=fz, x=x)
    return x, logp


# --------------------------------------------------------------------------------------
# Chamfer (pytorch3d.loss.chamfer_distance mean/mean, loss.py:42; kaolin form loss.py:35;
# nearest-neighbour rule from evaluation/tf_ops/nn_distance/tf_nndistance.cpp:21-43)
# --------------------------------------------------------------------------------------
def chamfer_nn(x: Tensor, y: Tensor):
    """dist1 [B,N], idx1, dist2 [B,M], idx2; squared L2; first minimum wins ties
    (tf_nndistance.cpp:33 `k==0 || d<best`)."""
    d = pairwise_sqdist(x.float(), y.float())
    d1, i1 = torch.sort(d, dim=2, stable=True)
    d2, i2 = torch.sort(d, dim=1, stable=True)
    return d1[:, :, 0], i1[:, :, 0], d2[:, 0, :], i2[:, 0, :]


def chamfer_distance_mean(x: Tensor, y: Tensor) -> Tensor:
    """pytorch3d chamfer_distance(batch_reduction='mean', point_reduction='mean')."""
    d1, _, d2, _ = chamfer_nn(x, y)
    return (d1.mean(dim=1) + d2.mean(dim=1)).mean()


def chamfer_distance_per_sample(x: Tensor, y: Tensor) -> Tensor:
    """kaolin-style per-sample mean+mean -> [B] (summed by ChamferCUDA2, loss.py:35-36)."""
    d1, _, d2, _ = chamfer_nn(x, y)
    return d1.mean(dim=1) + d2.mean(dim=1)


# --------------------------------------------------------------------------------------
# Training-mode restatement (differentiable; BN batch statistics, ActNorm data-dependent init)
# reference: train_pu1k.py:53-74 calls the same module in train() mode
# --------------------------------------------------------------------------------------
class TrainState:
    """Side effects of one train-mode forward: BN running-stat updates and ActNorm init values."""

    def __init__(self):
        self.bn_updates: Dict[str, Tensor] = {}
        self.bn_batch: Dict[str, Tuple[Tensor, Tensor]] = {}      # raw batch (mean, biased variance) per BN
        self.actnorm_init: Dict[str, Tensor] = {}


def _bn_train(sd: SD, pfx: str, y: Tensor, ts: TrainState) -> Tensor:
    """BatchNorm2d in training mode on [B,C,N,K]: biased batch variance for normalisation,
    unbiased for the running update, momentum 0.1."""
    out = F.batch_norm(y, None, None, sd[pfx + ".weight"], sd[pfx + ".bias"], training=True, eps=1e-5)
    with torch.no_grad():
        mean = y.mean(dim=(0, 2, 3))
        var_u = y.var(dim=(0, 2, 3), unbiased=True)
        ts.bn_batch[pfx] = (mean, y.var(dim=(0, 2, 3), unbiased=False))
        ts.bn_updates[pfx + ".running_mean"] = 0.9 * sd[pfx + ".running_mean"] + 0.1 * mean
        ts.bn_updates[pfx + ".running_var"] = 0.9 * sd[pfx + ".running_var"] + 0.1 * var_u
        ts.bn_updates[pfx + ".num_batches_tracked"] = sd[pfx + ".num_batches_tracked"] + 1
    return out


def edgeconv_unit_train(sd: SD, pfx: str, x: Tensor, idx: Tensor, ts: TrainState, pooling: bool = True) -> Tensor:
    nconv = 0
    while f"{pfx}.convs.{nconv}.0.weight" in sd:
        nconv += 1
    nb = knn_gather(x, idx)
    xi = x.unsqueeze(2).expand_as(nb)
    f = torch.cat([xi, nb, nb - xi], dim=-1).permute(0, 3, 1, 2)
    for t in range(nconv):
        y = F.conv2d(f, sd[f"{pfx}.convs.{t}.0.weight"], sd[f"{pfx}.convs.{t}.0.bias"])
        g = F.leaky_relu(_bn_train(sd, f"{pfx}.convs.{t}.1", y, ts), 0.05)
        f = torch.cat([f, g], dim=1)
    y = F.conv2d(f, sd[pfx + ".conv_out.weight"], sd[pfx + ".conv_out.bias"])
    if not pooling:
        return y
    return y.max(dim=-1)[0].transpose(1, 2)


def _actnorm_params(sd: SD, i: int, p: Tensor, ts: TrainState, init: bool):
    pf = f"flow_blocks.{i}.actnorm"
    logs, bias = sd[pf + ".logs"], sd[pf + ".bias"]
    if init:                                                    # normalize.py:45-54
        with torch.no_grad():
            b0 = -torch.mean(p.detach(), dim=(0, 1), keepdim=True)
            l0 = -torch.log(torch.std(p.detach(), dim=(0, 1), keepdim=True) + 1e-6)
            ts.actnorm_init[pf + ".bias"], ts.actnorm_init[pf + ".logs"] = b0, l0
        # the parameters keep their identity (copy_ into .data): gradients still flow to them
        logs = logs + (l0 - logs).detach()
        bias = bias + (b0 - bias).detach()
    return logs, bias


def forward_train(sd: SD, xyz: Tensor, upratio: int = 4, actnorm_init: bool = False):
    """Differentiable train-mode forward -> (x, logp, TrainState).  `sd` tensors may require grad."""
    ts = TrainState()
    xyz = xyz.float()
    with torch.no_grad():
        _, idx16 = knn_canonical(xyz, xyz, K_FEAT)
        idx8 = idx16[..., :K_INTERP].contiguous()
    # feature extractor
    cs, h = [], xyz
    for i in range(NUM_BLOCKS):
        h = edgeconv_unit_train(sd, f"feat_convs.{i}", h, idx16, ts)
        cs.append(feat_merge(sd, i, h))
    # f
    B, N, _ = xyz.shape
    p = xyz
    ldj = torch.zeros(B)
    an = []
    for i in range(NUM_BLOCKS):
        pf = f"flow_blocks.{i}"
        logs, bias = _actnorm_params(sd, i, p, ts, actnorm_init)
        an.append((logs, bias))
        p = p * torch.exp(logs) + bias
        ld = torch.sum(logs) * N
        W = sd[pf + ".permutate1.permutater.W"]
        p = torch.einsum("ij,bnj->bni", W, p)
        ld = ld + torch.slogdet(W)[1] * N
        h1, h2 = _split(p, i)
        h2 = h2 - lin_a1d(sd, pf + ".coupling1.bias_net", torch.cat([h1, cs[i]], dim=-1))
        p = torch.cat([h1, h2], dim=-1)[:, :, [2, 1, 0]]
        s = lin_a1d(sd, pf + ".coupling2.scale_net", cs[i])
        t = lin_a1d(sd, pf + ".coupling2.bias_net", cs[i])
        p = (p - t) * torch.exp(-s)
        ldj = ldj + ld - torch.sum(torch.flatten(s, start_dim=1), dim=1)
    z = p
    logp = -torch.mean(torch.sum(-0.5 * (z ** 2 + LOG2PI), dim=(1, 2)) + ldj)
    # interp (train-mode BN inside)
    nb = knn_gather(xyz, idx8)
    xi = xyz.unsqueeze(2).expand_as(nb)
    vec = xi - nb
    dist = torch.sqrt(torch.sum(vec ** 2, dim=-1, keepdim=True))
    fd = torch.cat([xi, nb, vec, dist], dim=-1).permute(0, 3, 1, 2)
    pd = "interp.knn_context.distance_encoder.mlp"
    d = fd
    for a in (0, 3):
        d = F.conv2d(d, sd[f"{pd}.{a}.weight"], sd[f"{pd}.{a}.bias"])
        d = F.leaky_relu(_bn_train(sd, f"{pd}.{a + 1}", d, ts), 0.01)
    d = F.conv2d(d, sd[pd + ".6.weight"], sd[pd + ".6.bias"])
    feat = edgeconv_unit_train(sd, "interp.knn_context.feat_conv", xyz, idx8, ts, pooling=False)
    w = torch.cat([d, feat], dim=1)
    pw = "interp.weight_unit.mlp"
    for a in (0, 3):
        w = F.conv2d(w, sd[f"{pw}.{a}.weight"], sd[f"{pw}.{a}.bias"])
        w = F.leaky_relu(_bn_train(sd, f"{pw}.{a + 1}", w, ts), 0.01)
    w = F.conv2d(w, sd[pw + ".6.weight"], sd[pw + ".6.bias"]).permute(0, 2, 1, 3)
    a_ = F.softmax(w[:, :, :upratio], dim=-1)
    fz = torch.einsum("bnck,bnrk->bncr", knn_gather(z, idx8).permute(0, 1, 3, 2), a_)
    # g
    u = torch.flatten(fz.transpose(2, 3), 1, 2)
    for i in reversed(range(NUM_BLOCKS)):
        pf = f"flow_blocks.{i}"
        c = torch.repeat_interleave(cs[i], upratio, dim=1)
        s = lin_a1d(sd, pf + ".coupling2.scale_net", c)
        t = lin_a1d(sd, pf + ".coupling2.bias_net", c)
        u = (u * torch.exp(s) + t)[:, :, [2, 1, 0]]
        h1, h2 = _split(u, i)
        h2 = h2 + lin_a1d(sd, pf + ".coupling1.bias_net", torch.cat([h1, c], dim=-1))
        u = torch.cat([h1, h2], dim=-1)
        u = torch.einsum("ij,bnj->bni", torch.inverse(sd[pf + ".permutate1.permutater.W"]), u)
        logs, bias = an[i]
        u = (u - bias) * torch.exp(-logs)
    return u, logp, ts


def calibrate(sd: SD, xyz: Tensor, upratio: int = 4) -> SD:
    """Make a synthetic "trained-style" state_dict self-consistent: BN running stats := the batch statistics
    of `xyz` (so eval == train on that batch), ActNorm := its data-dependent init.  Test infrastructure."""
    out = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        _, _, ts = forward_train(out, xyz, upratio, actnorm_init=True)
    for pfx, (m, v) in ts.bn_batch.items():
        out[pfx + ".running_mean"], out[pfx + ".running_var"] = m.clone(), v.clone()
    for k, v in ts.actnorm_init.items():
        out[k] = v.clone()
    return out
