#!/usr/bin/env python3
"""GPU box: pf_knn at the headline shape (32 x 2048, K = 16) - run it once per library build for an A/B
   PF_LIB_PATH=.../libpuflow_hip_knn4.so python tools/time_knn5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops
from puflow_amd.weights import synth_patches

for B, N, K in ((32, 2048, 16), (32, 2048, 8), (64, 1024, 16), (16, 2048, 16), (8, 2048, 16), (4, 2048, 16), (32, 256, 16), (2496, 256, 16), (32, 256, 8), (32, 1024, 16)):
    p = synth_patches(B, N, seed=1).cuda()
    for _ in range(3):
        ops.knn_idx32(p, p, K)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.knn_idx32(p, p, K)
    b.record(); torch.cuda.synchronize()
    print(f"knn {B} x {N}, K = {K}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us", flush=True)
