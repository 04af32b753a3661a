#!/usr/bin/env python3
"""Golden vectors for farthest point sampling from the REFERENCE's own in-tree implementation
(`modules/utils/fps.py::farthest_point_sampling`, the torch FPS `PatchHelper.merge_pc` uses, patch.py:162-165):
same algorithm as the un-vendored pointnet2 CUDA op (start at index 0, running min-distance initialised to 1e10,
squared distance summed x, y, z, arg-max).  Runs ONLY in the build container (needs /root/reference).
Writes tests/golden/fps_ref.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from modules.utils.fps import farthest_point_sampling, index_points      # noqa: E402  (the reference)
from puflow_amd.weights import synth_patches                             # noqa: E402


def main():
    out = {}
    for tag, (B, N, n, surface) in {"a": (2, 300, 50, False), "b": (1, 2048, 128, True), "c": (1, 700, 700, False),
                                    "d": (1, 10000, 400, True), "e": (3, 9000, 64, False)}.items():
        xyz = synth_patches(B, N, seed=100 + N, surface=surface)
        if tag == "c":
            xyz[0, 5] = xyz[0, 9]                       # duplicate points: ties
            xyz[0, N - 1] = xyz[0, 3]
        idx = farthest_point_sampling(xyz, n, RAN=True)
        out[f"{tag}_xyz"] = xyz.numpy()
        out[f"{tag}_idx"] = idx.numpy().astype(np.int64)
        print(tag, B, N, n, idx[0, :6].tolist())
    # merge_pc (patch.py:162-165): FPS over the concatenation, then gather
    a, b = synth_patches(2, 200, seed=1), synth_patches(2, 150, seed=2, surface=False)
    tmp = torch.cat([a, b], dim=1)
    out["m_a"], out["m_b"] = a.numpy(), b.numpy()
    out["m_out"] = index_points(tmp, farthest_point_sampling(tmp, 128)).numpy()
    path = os.path.join(ROOT, "tests", "golden", "fps_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
