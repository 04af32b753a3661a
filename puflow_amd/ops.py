"""Operator surface the reference model calls, backed by the HIP library.

Signatures mirror the third-party ops the reference imports:
  knn_points / knn_gather     pytorch3d.ops           (modules/discrete/interpflow.py:9,104,229,328)
  chamfer_distance            pytorch3d.loss          (metric/loss.py:14,42)
  history_chamfer_distance    kaolin.metrics.pointcloud (metric/loss.py:12,35)
All tensors must live on the GPU; there is no CPU path here.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.PuflowHipError("puflow_amd ops need GPU tensors (no CPU fallback)")
    return t.detach().contiguous().float()


def knn_idx32(p1: torch.Tensor, p2: torch.Tensor, K: int, want_dist: bool = False):
    """int32 device-format kNN: idx [B,N,K] (+ dists)."""
    lib = _lib.load()
    p1, p2 = _f32c(p1), _f32c(p2)
    B, N, _ = p1.shape
    M = p2.shape[1]
    idx = torch.empty((B, N, K), dtype=torch.int32, device=p1.device)
    dist = torch.empty((B, N, K), dtype=torch.float32, device=p1.device) if want_dist else None
    dptr = dist.data_ptr() if want_dist else None
    if K in (4, 8, 16, 32):
        _lib.check(lib.pf_knn(p1.data_ptr(), p2.data_ptr(), B, N, M, K, idx.data_ptr(), dptr, _stream()), "pf_knn")
    else:                       # any other K: sort-based kernel (same (distance, index) order), M <= 16384
        _lib.check(lib.pf_knn_large(p2.data_ptr(), p1.data_ptr(), B, M, N, K, idx.data_ptr(), dptr, _stream()), "pf_knn_large")
    return idx, dist


def _chamfer_bwd(lib):
    """pf_chamfer_bwd, or its atomics-free form under train_ops.set_deterministic(True) (a debugging switch)."""
    from . import train_ops
    return lib.pf_chamfer_bwd_det if train_ops.deterministic() else lib.pf_chamfer_bwd


def knn_points(p1: torch.Tensor, p2: torch.Tensor, K: int, return_nn: bool = False, return_sorted: bool = True,
               **_unused) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """(dists [B,N,K] squared L2, idx [B,N,K] int64, nn or None); always sorted by (dist, idx)."""
    idx, dist = knn_idx32(p1, p2, K, want_dist=True)
    idx64 = idx.long()
    nn = knn_gather(p2, idx64) if return_nn else None
    return dist, idx64, nn


def knn_gather(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,n,k,:] = x[b, idx[b,n,k], :]  (plain indexing; data movement only)."""
    B = x.shape[0]
    return x[torch.arange(B, device=x.device).view(B, 1, 1), idx.long()]


# ----------------------------------------------------------------------------------------
# Chamfer
# ----------------------------------------------------------------------------------------
def chamfer_nn(x: torch.Tensor, y: torch.Tensor):
    """dist1 [B,N], dist2 [B,M], idx1, idx2 (int32) + per-sample (mean+mean) [B] and {mean_b, sum_b} [2]."""
    lib = _lib.load()
    x, y = _f32c(x), _f32c(y)
    B, N, _ = x.shape
    M = y.shape[1]
    dev = x.device
    d1 = torch.empty((B, N), dtype=torch.float32, device=dev)
    d2 = torch.empty((B, M), dtype=torch.float32, device=dev)
    i1 = torch.empty((B, N), dtype=torch.int32, device=dev)
    i2 = torch.empty((B, M), dtype=torch.int32, device=dev)
    per = torch.empty((B,), dtype=torch.float32, device=dev)
    ms = torch.empty((2,), dtype=torch.float32, device=dev)
    _lib.check(lib.pf_chamfer_fwd(x.data_ptr(), y.data_ptr(), B, N, M, d1.data_ptr(), i1.data_ptr(), d2.data_ptr(),
                                  i2.data_ptr(), per.data_ptr(), ms.data_ptr(), _stream()), "pf_chamfer_fwd")
    return d1, d2, i1, i2, per, ms


class _ChamferFn(torch.autograd.Function):
    """outputs: per-sample chamfer [B] (mean over points of both directions)."""

    @staticmethod
    def forward(ctx, x, y):
        xc, yc = _f32c(x), _f32c(y)
        d1, d2, i1, i2, per, _ = chamfer_nn(xc, yc)
        ctx.save_for_backward(xc, yc, i1, i2)
        return per

    @staticmethod
    def backward(ctx, gper):
        x, y, i1, i2 = ctx.saved_tensors
        lib = _lib.load()
        B, N, _ = x.shape
        M = y.shape[1]
        gper = gper.contiguous().float()
        g1 = (gper / N).view(B, 1).expand(B, N).contiguous()
        g2 = (gper / M).view(B, 1).expand(B, M).contiguous()
        gx, gy = torch.zeros_like(x), torch.zeros_like(y)
        _lib.check(_chamfer_bwd(lib)(x.data_ptr(), y.data_ptr(), i1.data_ptr(), i2.data_ptr(), g1.data_ptr(),
                                      g2.data_ptr(), gx.data_ptr(), gy.data_ptr(), B, N, M, _stream()), "pf_chamfer_bwd")
        return gx, gy


def chamfer_distance(x, y, x_normals=None, y_normals=None, batch_reduction="mean", point_reduction="mean"):
    """pytorch3d.loss.chamfer_distance surface used by the reference (metric/loss.py:42): -> (loss, None)."""
    if x_normals is not None or y_normals is not None:
        raise NotImplementedError("normals are never passed on the PU-Flow path (train_pugan.py:60)")
    if point_reduction != "mean" or batch_reduction not in ("mean", "sum"):
        raise NotImplementedError("only the reductions the reference uses are built")
    per = _ChamferFn.apply(x, y)
    return (per.mean() if batch_reduction == "mean" else per.sum()), None


def history_chamfer_distance(p1, p2):
    """kaolin.metrics.pointcloud.chamfer_distance surface (metric/loss.py:35): per-sample [B]."""
    return _ChamferFn.apply(p1, p2)


def nearest_distance(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dist1 of chamfer_3DDist alone: squared distance of every point of x [B,N,3] to its nearest point of y [B,M,3] -> [B,N]
    (pf_nn1; PatchHelper.remove_outliers uses only this half, modules/utils/patch.py:199-203)."""
    lib = _lib.load()
    x, y = _f32c(x), _f32c(y)
    B, N, _ = x.shape
    d1 = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.check(lib.pf_nn1(x.data_ptr(), y.data_ptr(), B, N, y.shape[1], d1.data_ptr(), None, _stream()), "pf_nn1")
    return d1


class chamfer_3DDist:
    """ChamferDistancePytorch surface used by PatchHelper.remove_outliers (modules/utils/patch.py:199-203)."""

    def __call__(self, a, b):
        d1, d2, i1, i2, _, _ = chamfer_nn(a, b)
        return d1, d2, i1, i2


# ----------------------------------------------------------------------------------------
# Patch pipeline operators (pointnet2_ops / knn_cuda surfaces used by modules/utils/patch.py)
# ----------------------------------------------------------------------------------------
def furthest_point_sample(xyz: torch.Tensor, npoint: int, group: int = 0) -> torch.Tensor:
    """pointnet2_ops.pointnet2_utils.furthest_point_sample: xyz [B,N,3] -> int32 [B,npoint] (starts at index 0).
    group > 0: a layout hint, not a change of the result - every `group` consecutive points are one spatial neighbourhood
    (pf_fps_grouped: the cooperative kernel then gives a wave exactly one neighbourhood when it can)."""
    lib = _lib.load()
    xyz = _f32c(xyz)
    B, N, _ = xyz.shape
    idx = torch.zeros((B, npoint), dtype=torch.int32, device=xyz.device)
    mind = torch.empty((B, N), dtype=torch.float32, device=xyz.device)
    _lib.check(lib.pf_fps_grouped(xyz.data_ptr(), B, N, npoint, int(group), mind.data_ptr(), idx.data_ptr(), _stream()), "pf_fps")
    _check_fps_abort(lib, mind, B, N)
    return idx


def _check_fps_abort(lib, mind: torch.Tensor, B: int, N: int) -> None:
    """The cooperative FPS kernel's workgroups wait for each other with a BOUNDED spin.  A cloud's status word is 0 only when
    all of its steps completed (2 = never finished / never started, 1 = its workgroups gave up waiting): anything else means
    an invalid index row.  One small device -> host read per call (FPS itself is tens of ms)."""
    import ctypes
    stride, word = ctypes.c_longlong(0), ctypes.c_longlong(0)
    if not lib.pf_fps_scratch_layout(N, ctypes.byref(stride), ctypes.byref(word)):
        return
    words = mind.view(-1)[: (B * N) // 2 * 2].view(torch.int64)
    pos = torch.arange(B, device=mind.device, dtype=torch.int64) * stride.value + word.value
    if bool((words[pos] != 0).any()):
        raise _lib.PuflowHipError("pf_fps: the cooperative kernel did not complete every cloud (its workgroups timed out "
                                  "waiting for each other or were never co-resident); the sampled indices are invalid")


def normalize_pc(pc: torch.Tensor):
    """PatchHelper.normalize_pc (modules/utils/patch.py:168-178) on the GPU: pc [B,N,3] -> (normalised [B,N,3],
    centroid [B,1,3], furthest distance [B,1,1]).  Fixed summation order per cloud: independent of the batch size."""
    lib = _lib.load()
    pc = _f32c(pc)
    B, N, _ = pc.shape
    out = torch.empty_like(pc)
    cen = torch.empty((B, 1, 3), dtype=torch.float32, device=pc.device)
    fd = torch.empty((B, 1, 1), dtype=torch.float32, device=pc.device)
    _lib.check(lib.pf_normalize_pc(pc.data_ptr(), B, N, out.data_ptr(), cen.data_ptr(), fd.data_ptr(), _stream()),
               "pf_normalize_pc")
    return out, cen, fd


def gather_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """pointnet2 gather_operation: features [B,C,N], idx [B,M] -> [B,C,M] (indexing only)."""
    B, C, _ = features.shape
    return torch.gather(features, 2, idx.long().unsqueeze(1).expand(B, C, idx.shape[1]))


class KNN:
    """knn_cuda.KNN surface (modules/utils/patch.py:33,107): KNN(k, transpose_mode=False)(ref [B,C,N], query [B,C,M])
    -> (dist [B,k,M] squared L2, idx [B,k,M] int64), ordered by (distance, index)."""

    def __init__(self, k: int, transpose_mode: bool = False):
        self.k = k
        self.transpose_mode = transpose_mode

    def __call__(self, ref: torch.Tensor, query: torch.Tensor):
        lib = _lib.load()
        if not self.transpose_mode:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        ref, query = _f32c(ref), _f32c(query)
        B, N, _ = ref.shape
        M = query.shape[1]
        idx = torch.empty((B, M, self.k), dtype=torch.int32, device=ref.device)
        dist = torch.empty((B, M, self.k), dtype=torch.float32, device=ref.device)
        if self.k <= 32 and self.k in (4, 8, 16, 32):
            _lib.check(lib.pf_knn(query.data_ptr(), ref.data_ptr(), B, M, N, self.k, idx.data_ptr(), dist.data_ptr(),
                                  _stream()), "pf_knn")
        else:
            _lib.check(lib.pf_knn_large(ref.data_ptr(), query.data_ptr(), B, N, M, self.k, idx.data_ptr(), dist.data_ptr(),
                                        _stream()), "pf_knn_large")
        idx = idx.long()
        if not self.transpose_mode:
            return dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
        return dist, idx
