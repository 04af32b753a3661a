"""CPU: the patch-pipeline oracle against golden vectors from the REFERENCE's own in-tree torch FPS
(`modules/utils/fps.py`, tools/make_golden_patch.py) - pins the FPS semantics (start index, initial distance,
distance arithmetic, arg-max ties) that the un-vendored pointnet2 op leaves open."""
import os

import numpy as np
import torch

from oracle import patch_ref as P


def test_fps_oracle_matches_reference_torch_fps(golden_dir):
    g = np.load(os.path.join(golden_dir, "fps_ref.npz"))
    for tag in "abcde":
        xyz, ref = torch.from_numpy(g[f"{tag}_xyz"]), torch.from_numpy(g[f"{tag}_idx"])
        assert torch.equal(P.fps(xyz, ref.shape[1]).long(), ref), tag
