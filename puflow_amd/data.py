"""Patch datasets for the training entry points.

Mirrors what the reference's PU1K data path feeds `TrainerModule` (`dataset/pu1k/fetcher.py:11-48` loader and normalisation,
`:69-101` batch assembly and augmentation, `dataset/pu1k/dataset.py:24-52` dict batches) without TensorFlow / Lightning /
a background thread: a plain iterator of batches that are already torch tensors on the training device.

File formats: the reference's HDF5 container (datasets `poisson_<n>`; needs `h5py`, which this image does not ship: the
reader raises with that message instead of failing at import) and an `.npz` with the same array names (tests, synthetic data).
`SyntheticPatchData` makes surface patches in memory (the build / bench machines have no datasets).
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch


def load_patch_arrays(path: str, num_point: int = 256, up_ratio: int = 4, use_random_input: bool = False,
                      skip_rate: int = 1) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (input [M, n_in, 3], gt [M, num_point * up_ratio, 3], radius [M]) fp32, normalised like the reference
    (`fetcher.py:32-40`): both clouds are centred on the INPUT's centroid and divided by the input's furthest distance."""
    n_in = num_point * 4 if use_random_input else num_point
    n_out = num_point * up_ratio
    if path.endswith(".npz"):
        with np.load(path) as f:
            inp, gt = f[f"poisson_{n_in}"].astype(np.float32), f[f"poisson_{n_out}"].astype(np.float32)
    else:
        try:
            import h5py
        except ImportError as e:               # pragma: no cover - h5py is not installed in the build image
            raise ImportError("reading the reference's .h5 patch files needs h5py (not installed); convert the two arrays "
                              f"'poisson_{n_in}' / 'poisson_{n_out}' to an .npz with the same names") from e
        with h5py.File(path, "r") as f:       # pragma: no cover
            inp, gt = f[f"poisson_{n_in}"][:].astype(np.float32), f[f"poisson_{n_out}"][:].astype(np.float32)
    if len(inp) != len(gt):
        raise ValueError("input / ground-truth patch counts differ")
    inp, gt = inp[..., :3].copy(), gt[..., :3].copy()
    centroid = inp.mean(axis=1, keepdims=True)
    inp -= centroid
    far = np.sqrt((inp ** 2).sum(-1)).max(axis=1, keepdims=True)[..., None]
    inp /= far
    gt = (gt - centroid) / far
    radius = np.ones(len(inp), np.float32)
    return inp[::skip_rate], gt[::skip_rate], radius[::skip_rate]


# ---- augmentation (the reference applies jitter -> rotation -> scale, fetcher.py:96-100) --------------------------------
def jitter(rng: np.random.Generator, x: np.ndarray, sigma: float, clip: float) -> np.ndarray:
    return x + np.clip(sigma * rng.standard_normal(x.shape), -clip, clip).astype(np.float32)


def random_rotations(rng: np.random.Generator, n: int) -> np.ndarray:
    """[n, 3, 3] rotations Rz Ry Rx with independent uniform angles (applied as row-vector x @ R)."""
    a = rng.uniform(0.0, 2.0 * np.pi, size=(n, 3))
    c, s = np.cos(a), np.sin(a)
    R = np.zeros((n, 3, 3, 3))
    R[:, 0] = np.eye(3); R[:, 1] = np.eye(3); R[:, 2] = np.eye(3)
    R[:, 0, 1, 1], R[:, 0, 1, 2], R[:, 0, 2, 1], R[:, 0, 2, 2] = c[:, 0], -s[:, 0], s[:, 0], c[:, 0]        # about x
    R[:, 1, 0, 0], R[:, 1, 0, 2], R[:, 1, 2, 0], R[:, 1, 2, 2] = c[:, 1], s[:, 1], -s[:, 1], c[:, 1]        # about y
    R[:, 2, 0, 0], R[:, 2, 0, 1], R[:, 2, 1, 0], R[:, 2, 1, 1] = c[:, 2], -s[:, 2], s[:, 2], c[:, 2]        # about z
    return (R[:, 2] @ R[:, 1] @ R[:, 0]).astype(np.float32)


def augment(rng: np.random.Generator, inp: np.ndarray, gt: np.ndarray, radius: np.ndarray, jitter_sigma: float,
            jitter_max: float, scale_low: float = 0.8, scale_high: float = 1.2):
    inp = jitter(rng, inp, jitter_sigma, jitter_max)
    R = random_rotations(rng, len(inp))
    inp, gt = np.einsum("bnk,bkl->bnl", inp, R), np.einsum("bnk,bkl->bnl", gt, R)
    sc = rng.uniform(scale_low, scale_high, size=len(inp)).astype(np.float32)
    return inp * sc[:, None, None], gt * sc[:, None, None], radius * sc


class PatchData:
    """Iterable of dict batches with the reference's keys (`dataset.py:45-52`): 'input_sparse_xyz_pl' [B, n, 3],
    'gt_dense_xyz_pl' [B, n * up_ratio, 3], 'up_ratio_pl' [B] (the patch radius).  One pass = `num_batches` batches of
    a fresh shuffle; under torch.distributed every rank draws the same shuffle and keeps its contiguous shard."""

    def __init__(self, inp: np.ndarray, gt: np.ndarray, radius: Optional[np.ndarray] = None, batch_size: int = 32,
                 num_point_patch: int = 256, use_random_input: bool = False, is_augment: bool = True,
                 jitter_sigma: float = 0.01, jitter_max: float = 0.03, num_batches: Optional[int] = None,
                 device: str = "cpu", seed: int = 2021, rank: int = 0, world: int = 1):
        self.inp, self.gt = inp, gt
        self.radius = radius if radius is not None else np.ones(len(inp), np.float32)
        self.batch_size, self.npoint, self.random_input = batch_size, num_point_patch, use_random_input
        self.is_augment, self.jitter_sigma, self.jitter_max = is_augment, jitter_sigma, jitter_max
        self.num_batches = num_batches if num_batches is not None else len(inp) // batch_size
        self.device, self.rank, self.world = device, rank, world
        self.rng = np.random.default_rng(seed)                      # the same stream on every rank

    def __len__(self) -> int:
        return self.num_batches

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        from .dist import shard_bounds
        order = self.rng.permutation(len(self.inp))
        for b in range(self.num_batches):
            sel = order[(b * self.batch_size) % len(order):][:self.batch_size]
            if len(sel) < self.batch_size:                           # wrap (num_batches may exceed one pass)
                sel = np.concatenate([sel, order[:self.batch_size - len(sel)]])
            inp, gt, rad = self.inp[sel].copy(), self.gt[sel].copy(), self.radius[sel].copy()
            if self.random_input:                                    # non-uniform subsample of the 4x input (fetcher.py:88-95)
                new = np.zeros((len(sel), self.npoint, 3), np.float32)
                for i in range(len(sel)):
                    new[i] = inp[i][self._nonuniform(inp.shape[1], self.npoint)]
                inp = new
            if self.is_augment:
                inp, gt, rad = augment(self.rng, inp, gt, rad, self.jitter_sigma, self.jitter_max)
            lo, hi = shard_bounds(len(sel), self.rank, self.world)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi], dtype=np.float32)).to(self.device)
            yield {"input_sparse_xyz_pl": t(inp), "gt_dense_xyz_pl": t(gt), "up_ratio_pl": t(rad)}

    def _nonuniform(self, num: int, sample_num: int) -> np.ndarray:
        """Indices clustered around a random location (Gaussian in index space), without repetition."""
        loc = self.rng.uniform(0.1, 0.9)
        chosen: Dict[int, None] = {}
        while len(chosen) < sample_num:
            for a in (self.rng.normal(loc, 0.3, size=2 * sample_num) * num).astype(np.int64):
                if 0 <= a < num and len(chosen) < sample_num:
                    chosen.setdefault(int(a))
        return np.fromiter(chosen, dtype=np.int64)


def patch_data_from_file(path: str, **kw) -> PatchData:
    npoint, up = kw.get("num_point_patch", 256), kw.pop("up_ratio", 4)
    inp, gt, rad = load_patch_arrays(path, npoint, up, kw.get("use_random_input", False))
    return PatchData(inp, gt, rad, **kw)


class SyntheticPatchData(PatchData):
    """Surface patches made in memory (`weights.synth_patches`): dense = n*up points on a random smooth surface, sparse =
    a subset of them, both normalised by the SPARSE cloud like `load_patch_arrays`."""

    def __init__(self, num_patches: int = 256, num_point_patch: int = 256, up_ratio: int = 4, seed: int = 2021, **kw):
        from .weights import synth_patches
        dense = synth_patches(num_patches, num_point_patch * up_ratio, seed=seed).numpy()
        rng = np.random.default_rng(seed + 1)
        sparse = np.stack([d[rng.permutation(d.shape[0])[:num_point_patch]] for d in dense])
        c = sparse.mean(axis=1, keepdims=True)
        far = np.sqrt(((sparse - c) ** 2).sum(-1)).max(axis=1, keepdims=True)[..., None]
        super().__init__(((sparse - c) / far).astype(np.float32), ((dense - c) / far).astype(np.float32),
                         num_point_patch=num_point_patch, seed=seed, **kw)
