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
    _lib.check(lib.pf_knn(p1.data_ptr(), p2.data_ptr(), B, N, M, K, idx.data_ptr(),
                          dist.data_ptr() if want_dist else None, _stream()), "pf_knn")
    return idx, dist


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
