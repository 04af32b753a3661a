#!/usr/bin/env python3
"""GPU box: 30 launches of the brute-force kNN at the bench shape (32 x 2048, K = 16) and of the Chamfer nearest-neighbour
search at 32 x 8192 - a target for the profiler scripts (bash tools/pmc_sq.sh r2_knn tools/run_knn.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops
from puflow_amd.weights import synth_patches

xyz = synth_patches(32, 2048, seed=2021).cuda()
big = synth_patches(32, 8192, seed=7).cuda()
for _ in range(30):
    ops.knn_idx32(xyz, xyz, 16)
for _ in range(5):
    ops.nearest_distance(big, big)
torch.cuda.synchronize()
print("done")
