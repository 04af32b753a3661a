"""CPU oracle for the patch pipeline (FPS, large-K kNN, PatchHelper).  TEST INFRASTRUCTURE ONLY.

Restates `modules/utils/patch.py:35-214` step by step.  The kernels it calls in the reference come from
un-vendored, unpinned third-party packages (pointnet2_ops FPS, knn_cuda KNN, ChamferDistancePytorch;
docker/Dockerfile:47-49, .gitmodules:1-3) and the reference has no tests for them.
  * FPS: PINNED to the reference's own in-tree torch implementation of the same algorithm
    (`modules/utils/fps.py::farthest_point_sampling`, used by `PatchHelper.merge_pc`, patch.py:162-165):
    tests/golden/fps_ref.npz (tools/make_golden_patch.py) fixes the start index (0), the 1e10 initial distance, the
    (dx^2 + dy^2) + dz^2 arithmetic and the first-maximum tie rule; `fps` below and `pf_fps` reproduce it exactly.
  * large-K kNN, Chamfer-based outlier removal: **parity unpinned** - the semantics are the ones stated in
    csrc/patch_ops.hip (kNN by (distance, index); unfused fp32 squared distances)."""
from __future__ import annotations

import torch

from . import ref_cpu as O

Tensor = torch.Tensor


def fps(xyz: Tensor, npoint: int) -> Tensor:
    """[B,N,3] -> int64 [B,npoint].  numpy on coordinate planes with preallocated temporaries (20 024 steps over 99 840
    points in seconds); the arithmetic and its order are the pinned ones: fp32 (dx^2 + dy^2) + dz^2, running minimum,
    first maximal value."""
    import numpy as np
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), np.int64)
    for b in range(B):
        p = np.ascontiguousarray(xyz[b].detach().cpu().float().numpy())
        px, py, pz = p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()
        mind = np.full(N, 1e10, np.float32)
        t0, t1 = np.empty(N, np.float32), np.empty(N, np.float32)
        cur = 0
        for j in range(1, npoint):
            np.subtract(px, px[cur], out=t0); np.multiply(t0, t0, out=t0)
            np.subtract(py, py[cur], out=t1); np.multiply(t1, t1, out=t1)
            np.add(t0, t1, out=t0)
            np.subtract(pz, pz[cur], out=t1); np.multiply(t1, t1, out=t1)
            np.add(t0, t1, out=t0)
            np.minimum(mind, t0, out=mind)
            cur = int(np.argmax(mind))               # first maximal value
            out[b, j] = cur
    return torch.from_numpy(out)


def normalize_pc(pc: Tensor):
    centroid = torch.mean(pc, dim=1, keepdim=True)
    pc = pc - centroid
    fd, _ = torch.max(torch.sum(pc ** 2, dim=-1, keepdim=True).sqrt(), dim=1, keepdim=True)
    return pc / fd, centroid, fd


def extract_knn_patch(pc: Tensor, npoint_patch: int, expand_ratio: float) -> Tensor:
    B, N, C = pc.shape
    n_patch = int(N / npoint_patch * expand_ratio)
    seeds = fps(pc, n_patch)
    cent = pc[torch.arange(B).view(B, 1), seeds]                    # [B,n_patch,3]
    _, idx = O.knn_canonical(cent, pc, npoint_patch)                # [B,n_patch,k]
    return pc[torch.arange(B).view(B, 1, 1), idx]                   # [B,n_patch,k,3]


def merge_patches(patches: Tensor, npoint: int) -> Tensor:
    B, _, _, C = patches.shape
    flat = patches.reshape(B, -1, C)
    idx = fps(flat, npoint)
    return flat[torch.arange(B).view(B, 1), idx]                    # [B,npoint,3]


def remove_outliers(sr: Tensor, lr: Tensor, num_outliers: int) -> Tensor:
    d1, _, _, _ = O.chamfer_nn(sr, lr)
    B, N = d1.shape
    order = torch.argsort(d1, dim=-1, descending=True, stable=True)[:, :num_outliers]
    keep = torch.ones(B, N, dtype=torch.bool)
    keep[torch.arange(B).view(B, 1), order] = False
    return torch.stack([sr[b][keep[b]] for b in range(B)])
