"""Inference CLI - same flags, defaults, file handling and output format as the reference's
`modules/discrete/upsample.py:20-86`:

  python -m puflow_amd.upsample --source=path/to/input --target=path/to/output --checkpoint=ckpt.pt --up_ratio=4
"""
from __future__ import annotations

import os
from argparse import ArgumentParser
from pathlib import Path
from typing import List

import numpy as np
import torch

from .interpflow import PointInterpFlow
from .patch import PatchHelper


def shard_paths(paths: List[str], rank: int, world: int) -> List[str]:
    """Files of rank `rank` out of `world`: sorted, then strided (balanced for any count)."""
    return sorted(paths)[rank::world] if world > 1 else list(paths)


def save_xyz(path, points: np.ndarray) -> None:
    """`np.savetxt(path, points, fmt="%.6f")` (upsample.py:57) - the same bytes - formatted by the library's host-side
    `pf_format_xyz` (~1 ms for a 20 000-point cloud instead of 36 ms; the call releases the GIL, so the writer thread really
    runs beside the GPU work).  With the GPU part batched, writing was what the CLI waited for."""
    points = np.atleast_2d(np.asarray(points))
    if points.dtype != np.float32:                       # the library formats float32 values; anything else: one format call
        line = " ".join(["%.6f"] * points.shape[1]) + "\n"
        with open(path, "w") as f:
            f.write((line * points.shape[0]) % tuple(points.ravel().tolist()))
        return
    import ctypes
    from . import _lib
    lib = _lib.load()
    pts = np.ascontiguousarray(points)
    n, c = pts.shape
    cap = lib.pf_format_xyz_bound(n, c)
    buf = ctypes.create_string_buffer(cap)
    m = lib.pf_format_xyz(pts.ctypes.data, n, c, ctypes.addressof(buf), cap)
    if m < 0:
        _lib.check(int(m), "pf_format_xyz")
    with open(path, "wb") as f:
        f.write(memoryview(buf)[:m])


def load_xyz(path) -> np.ndarray:
    """`np.loadtxt(path, dtype=np.float32)` (upsample.py:42) through the library's host-side parser (`pf_parse_xyz`: the same
    values, ~0.3 ms instead of 2-3 ms per 5000-point file); anything the parser does not take (a token that is not a plain
    number, other delimiters) goes to numpy."""
    import ctypes
    from . import _lib
    with open(path, "rb") as f:
        raw = f.read()
    lib = _lib.load()
    buf = ctypes.create_string_buffer(raw, len(raw) + 1)          # + terminating 0
    cap = len(raw) // 2 + 1                                       # a value takes at least one character and a separator
    out = np.empty(cap, dtype=np.float32)
    ncols = ctypes.c_int(0)
    n = lib.pf_parse_xyz(ctypes.addressof(buf), len(raw), out.ctypes.data, cap, ctypes.byref(ncols))
    if n <= 0 or ncols.value <= 0:
        return np.loadtxt(path, dtype=np.float32)                 # unusual file: numpy decides (and raises its own errors)
    arr = out[:n].reshape(-1, ncols.value)
    return arr[0].copy() if arr.shape[0] == 1 else (arr[:, 0].copy() if ncols.value == 1 else arr.copy())


@torch.no_grad()
def upsampling(data_paths: List[str], target_path: str, checkpoint_path: str, up_ratio: int, num_outlier: int,
               num_patch: int, num_upsampling: int = None, seed=None, state_dict=None, network_cls=PointInterpFlow,
               cloud_batch: int = None):
    # The continuous model integrates all patches of a pass with ONE adaptive step sequence (the error norm is taken over the
    # whole batch, cnf.py:97-113), so a cloud's result depends on what shares its pass: it keeps the reference's one file at a
    # time unless asked otherwise.  The discrete model's patches never interact.
    if cloud_batch is None:
        cloud_batch = 16 if network_cls is PointInterpFlow else 1
    # The host side of this loop is a 5000-element permutation and two small copies per file: ONE intra-op thread.  torch's
    # default is one thread per visible CPU (128 on the MI355X boxes, whose containers own 16 CPUs' worth of quota): every tiny
    # CPU op then wakes a pool of spinning OpenMP threads, the cgroup's CPU quota is gone within a few ms of each 100 ms period
    # and the WHOLE process - the thread waiting for the GPU included - is throttled for the rest of it: a third of the clouds took
    # 70 ms instead of 11 (tools/fps_interplay_probe.py shuffle [1]; round 5).
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _upsampling(data_paths, target_path, checkpoint_path, up_ratio, num_outlier, num_patch, num_upsampling, seed,
                           state_dict, network_cls, cloud_batch)
    finally:
        torch.set_num_threads(host_threads)


def _upsampling(data_paths, target_path, checkpoint_path, up_ratio, num_outlier, num_patch, num_upsampling, seed, state_dict,
                network_cls, cloud_batch):
    if seed is not None:
        np.random.seed(seed)
        torch.random.manual_seed(seed)
        torch.cuda.manual_seed(seed)
    # One process per GPU (torchrun): every rank upsamples its own files - clouds are independent, so the CLI shards
    # by file with no collective at all (BASELINE configs[3]); a plain `python -m puflow_amd.upsample` is rank 0 of 1.
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # The per-file shuffle is drawn from the process RNG in file order (upsample.py:43), so a file's result depends on the
    # files before it.  Under sharding every rank therefore walks the WHOLE sorted list and draws (and discards) the shuffles
    # of the other ranks' files: each file gets exactly the result of a one-process run, whatever the number of ranks.  The
    # list is sorted in every case (the reference takes os.walk's order, which depends on the file system: its outputs are
    # reproducible only on one machine).
    all_paths = sorted(data_paths)
    mine = set(shard_paths(data_paths, rank, world))
    device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)}")
    network = network_cls(3)
    network.load_state_dict(state_dict if state_dict is not None else torch.load(checkpoint_path, map_location="cpu"))
    network.set_to_initialized_state()
    network = network.to(device).eval()
    patch_helper = PatchHelper(num_patch, patch_expand_ratio=4)
    # The reference takes one file at a time (upsample.py:42-57).  Here up to `cloud_batch` consecutive files of the same
    # point count go through the pipeline together: the FPS merge, which is sequential in its 4N output points and 95 % of a
    # cloud's GPU time, then runs for all of them at once (16 clouds take the time of 1.35).  Every cloud's result is the one it
    # gets alone (clouds never interact; the per-file shuffles are drawn in file order as before), and finished clouds are
    # written by a worker thread while the GPU works on the next batch.
    from concurrent.futures import ThreadPoolExecutor
    writes = []
    pending = []                                                   # (file name, shuffled cloud [1,N,3] on the host)

    def flush(pool):
        if not pending:
            return
        pt_input = torch.cat([c for _, c in pending], dim=0).to(device)
        if num_upsampling is None:
            npoint = pt_input.shape[1] * up_ratio + (num_outlier or 0)
        else:
            npoint = num_upsampling + (num_outlier or 0)
        pred = patch_helper.upsample(network, pt_input, npoint=npoint, upratio=up_ratio, jitter=False)
        if num_outlier is not None and num_outlier > 0:
            pred = PatchHelper.remove_outliers(pred, pt_input, num_outlier)
        pred = pred.cpu().numpy()
        for (file_name, _), cloud in zip(pending, pred):
            writes.append(pool.submit(save_xyz, Path(target_path) / file_name, cloud))
        pending.clear()

    with ThreadPoolExecutor(max_workers=1) as pool:
        for path in all_paths:
            _, file_name = os.path.split(path)
            pt_input = torch.from_numpy(load_xyz(path)).unsqueeze(0)
            perm = torch.randperm(pt_input.shape[1])
            if path not in mine:
                continue                                            # another rank's file: only the RNG draw is replayed
            pt_input = pt_input[:, perm].contiguous()
            if pending and (pending[0][1].shape[1] != pt_input.shape[1] or len(pending) >= max(int(cloud_batch), 1)):
                flush(pool)
            pending.append((file_name, pt_input))
        flush(pool)
        for w in writes:
            w.result()                                             # re-raises a failed write


def main(argv=None, network_cls=PointInterpFlow):
    parser = ArgumentParser()
    parser.add_argument("--source", type=str, help="Path of input directory")
    parser.add_argument("--target", type=str, help="Path of output directory")
    parser.add_argument("--seed", type=int, default=2021)
    parser.add_argument("--checkpoint", type=str, help="Path of checkpoint")
    parser.add_argument("--up_ratio", type=int, help="upsampling ratio", default=4)
    parser.add_argument("--num_patch", type=int, help="number of point in each patch", default=256)
    parser.add_argument("--num_out", type=int, default=None, help="number of point of output point cloud")
    parser.add_argument("--cloud_batch", type=int, default=None,
                        help="(not in the reference) files of equal point count that share one pass; 1 = one file at a time "
                             "(default: 16 for the discrete model, 1 for the continuous one)")
    args = parser.parse_args(argv)
    os.makedirs(args.target, exist_ok=True)            # exist_ok: several ranks may race to create it
    data_paths = []
    for root, _dirs, files in os.walk(args.source):
        data_paths.extend([os.path.join(root, f) for f in files if ".xyz" in f])
    upsampling(data_paths, args.target, args.checkpoint, up_ratio=args.up_ratio, num_outlier=24, num_patch=args.num_patch,
               num_upsampling=args.num_out, seed=args.seed, network_cls=network_cls, cloud_batch=args.cloud_batch)


if __name__ == "__main__":
    main()
