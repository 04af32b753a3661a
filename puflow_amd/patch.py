"""PatchHelper: the patch pipeline around the network - normalise, FPS seeds, K=256 kNN patches, network,
concatenate, FPS merge, de-normalise, outlier removal.  Mirrors the reference's surface and step order
(`modules/utils/patch.py:18-214`); FPS / kNN / Chamfer run in the HIP library, the rest is tensor re-layout
and O(points x 3) normalisation arithmetic in torch, as in the reference.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops


class PatchHelper(object):

    def __init__(self, npoint_patch: int, patch_expand_ratio: float, extract: str = "knn"):
        if extract != "knn":
            raise NotImplementedError("only kNN patch extraction exists (as in the reference)")
        self._npoint_patch = npoint_patch
        self._patch_expand_ratio = patch_expand_ratio
        self.knn = ops.KNN(k=npoint_patch, transpose_mode=False)

    # ---- patch.py:35-80
    def upsample(self, upsampler, pc: Tensor, npoint: int, upratio=None, jitter: bool = False, **kwargs) -> Tensor:
        B, N, C = pc.shape
        pc, g_centroid, g_furthest_distance = PatchHelper.normalize_pc(pc)
        if jitter:
            pc = PatchHelper.jitter_perturbation_point_cloud(pc)
        patches = PatchHelper.extract_knn_patch(pc, self.knn, self._npoint_patch, self._patch_expand_ratio)
        patches = patches.reshape(B, -1, self._npoint_patch, C)
        predict_patches = PatchHelper.upsampling_patches(upsampler, patches, upratio, **kwargs)
        predict_pc = PatchHelper.merge_patches(predict_patches, npoint)            # [B,3,npoint]
        predict_pc = predict_pc * g_furthest_distance + g_centroid.transpose(1, 2)
        return predict_pc.transpose(1, 2).contiguous()

    # ---- patch.py:82-93
    @staticmethod
    def upsampling_patches(upsampler, patches: Tensor, upratio=None, **kwargs) -> Tensor:
        B, n_patch, k1, C = patches.shape
        patches = patches.reshape(B * n_patch, k1, C)
        patches, centroids, furthest_distance = PatchHelper.normalize_pc(patches)
        predict = upsampler.sample(patches.contiguous(), upratio=(upratio or 4), **kwargs)
        predict = torch.cat([predict, patches], dim=1)
        predict = predict * furthest_distance + centroids
        return predict.reshape(B, n_patch, -1, C)

    # ---- patch.py:95-125
    @staticmethod
    def extract_idx_patches(pc: Tensor, knn_searcher, npoint_patch: int, expand_ratio: float, seed_centroids_idx=None):
        _, N, _ = pc.shape
        pc_T = pc.transpose(1, 2).contiguous()
        if seed_centroids_idx is None:
            n_patch = int(N / npoint_patch * expand_ratio)
            patch_centroids_idx = ops.furthest_point_sample(pc, n_patch)
        else:
            n_patch = seed_centroids_idx.shape[1]
            patch_centroids_idx = seed_centroids_idx
        patch_centroids = ops.gather_operation(pc_T, patch_centroids_idx)          # [B,C,n_patch]
        _, idx_patches = knn_searcher(pc_T, patch_centroids)                       # [B,k,n_patch]
        return idx_patches, n_patch

    @staticmethod
    def extract_knn_patch(pc: Tensor, knn_searcher, npoint_patch: int, expand_ratio: float, seed_centroids_idx=None) -> Tensor:
        B, _, C = pc.shape
        idx_b = torch.arange(B, device=pc.device).view(-1, 1)
        idx_patches, n_patch = PatchHelper.extract_idx_patches(pc, knn_searcher, npoint_patch, expand_ratio, seed_centroids_idx)
        idx_patches = idx_patches.transpose(1, 2).flatten(start_dim=1)
        patches = pc[idx_b, idx_patches]
        return patches.reshape(B, n_patch, npoint_patch, C)

    @staticmethod
    def fps(pc: Tensor, n_point: int, transpose: bool = True) -> Tensor:
        idx = ops.furthest_point_sample(pc, n_point)
        cent = ops.gather_operation(pc.transpose(1, 2).contiguous(), idx)
        return cent.transpose(1, 2).contiguous() if transpose else cent

    # ---- patch.py:142-158
    @staticmethod
    def merge_patches(patches: Tensor, npoint: int, origins: Tensor = None) -> Tensor:
        B, _, per_patch, C = patches.shape
        patches = patches.reshape(B, -1, C)
        if origins is not None:
            patches = torch.cat([patches, origins], dim=1)
        patches = patches.contiguous()
        # layout hint (same indices, bit for bit): the candidates arrive patch after patch, `per_patch` consecutive points each
        idx = ops.furthest_point_sample(patches, npoint, group=per_patch if origins is None else 0)
        return ops.gather_operation(patches.transpose(1, 2).contiguous(), idx)     # [B,3,npoint]

    # ---- patch.py:168-178
    # ---- patch.py:162-165 (the reference uses its in-tree torch FPS here; same algorithm, see tests/golden/fps_ref.npz)
    @staticmethod
    def merge_pc(pc1: Tensor, pc2: Tensor, npoint: int) -> Tensor:
        tmp = torch.cat([pc1, pc2], dim=1).contiguous()
        idx = ops.furthest_point_sample(tmp, npoint).long()
        return tmp[torch.arange(tmp.shape[0], device=tmp.device).view(-1, 1), idx]

    @staticmethod
    def normalize_pc(pc: Tensor):
        if pc.is_cuda and pc.dim() == 3 and pc.shape[-1] == 3:
            return ops.normalize_pc(pc)          # fixed summation order: a cloud's result does not depend on the batch
        centroid = torch.mean(pc, dim=1, keepdim=True)
        pc = pc - centroid
        dist = torch.sum(pc ** 2, dim=-1, keepdim=True).sqrt()
        furthest_distance, _ = torch.max(dist, dim=1, keepdim=True)
        return pc / furthest_distance, centroid, furthest_distance

    @staticmethod
    def jitter_perturbation_point_cloud(pc: Tensor, sigma: float = 0.010, clip: float = 0.020) -> Tensor:
        if sigma <= 0:
            return pc
        B, N, C = pc.shape
        jit = torch.clamp(sigma * torch.randn(B, N, C, device=pc.device), -clip, clip)
        jit[:, :, 3:] = 0
        return jit + pc

    # ---- patch.py:198-214
    @staticmethod
    def remove_outliers(sr: Tensor, lr: Tensor, num_outliers: int) -> Tensor:
        (B, N, _), device = sr.shape, sr.device
        dist1 = ops.nearest_distance(sr, lr)       # = chamfer_3DDist()(sr, lr)[0] (patch.py:203 uses only dist1): half the work
        idx_outliers = torch.argsort(dist1, dim=-1, descending=True, stable=True)[:, :num_outliers]
        idxb = torch.arange(B, device=device).view(-1, 1)
        keep = torch.ones((B, N), dtype=torch.int32, device=device)
        keep[idxb, idx_outliers] = 0
        idx_inverse = torch.nonzero(keep, as_tuple=False)[:, 1].view(B, N - num_outliers)
        return sr[idxb, idx_inverse]
