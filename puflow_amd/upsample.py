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


@torch.no_grad()
def upsampling(data_paths: List[str], target_path: str, checkpoint_path: str, up_ratio: int, num_outlier: int,
               num_patch: int, num_upsampling: int = None, seed=None, state_dict=None, network_cls=PointInterpFlow):
    if seed is not None:
        np.random.seed(seed)
        torch.random.manual_seed(seed)
        torch.cuda.manual_seed(seed)
    # One process per GPU (torchrun): every rank upsamples its own files - clouds are independent, so the CLI shards
    # by file with no collective at all (BASELINE configs[3]); a plain `python -m puflow_amd.upsample` is rank 0 of 1.
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    data_paths = shard_paths(data_paths, rank, world)
    device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)}")
    network = network_cls(3)
    network.load_state_dict(state_dict if state_dict is not None else torch.load(checkpoint_path, map_location="cpu"))
    network.set_to_initialized_state()
    network = network.to(device).eval()
    patch_helper = PatchHelper(num_patch, patch_expand_ratio=4)
    for path in data_paths:
        _, file_name = os.path.split(path)
        pt_input = torch.from_numpy(np.loadtxt(path, dtype=np.float32)).unsqueeze(0).to(device)
        pt_input = pt_input[:, torch.randperm(pt_input.shape[1])].contiguous()
        if num_upsampling is None:
            npoint = pt_input.shape[1] * up_ratio + (num_outlier or 0)
        else:
            npoint = num_upsampling + (num_outlier or 0)
        pred = patch_helper.upsample(network, pt_input, npoint=npoint, upratio=up_ratio, jitter=False)
        if num_outlier is not None and num_outlier > 0:
            pred = PatchHelper.remove_outliers(pred, pt_input, num_outlier)
        np.savetxt(Path(target_path) / file_name, pred.squeeze().cpu().numpy(), fmt="%.6f")


def main(argv=None, network_cls=PointInterpFlow):
    parser = ArgumentParser()
    parser.add_argument("--source", type=str, help="Path of input directory")
    parser.add_argument("--target", type=str, help="Path of output directory")
    parser.add_argument("--seed", type=int, default=2021)
    parser.add_argument("--checkpoint", type=str, help="Path of checkpoint")
    parser.add_argument("--up_ratio", type=int, help="upsampling ratio", default=4)
    parser.add_argument("--num_patch", type=int, help="number of point in each patch", default=256)
    parser.add_argument("--num_out", type=int, default=None, help="number of point of output point cloud")
    args = parser.parse_args(argv)
    os.makedirs(args.target, exist_ok=True)            # exist_ok: several ranks may race to create it
    data_paths = []
    for root, _dirs, files in os.walk(args.source):
        data_paths.extend([os.path.join(root, f) for f in files if ".xyz" in f])
    upsampling(data_paths, args.target, args.checkpoint, up_ratio=args.up_ratio, num_outlier=24, num_patch=args.num_patch,
               num_upsampling=args.num_out, seed=args.seed, network_cls=network_cls)


if __name__ == "__main__":
    main()
