#!/usr/bin/env python3
"""GPU box: pf_nn1 (Chamfer's nearest neighbour) at the training shape and at PU-GAN's outlier-removal shape.
   PF_LIB_PATH=<variant .so> python tools/time_nn1.py  for an A/B (e.g. a -DPF_KNN5=0 build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.weights import synth_patches

lib = _lib.load()
for B, N in ((32, 1024), (32, 256), (8, 4096), (2, 20024)):
    x = synth_patches(B, N, seed=1).cuda(); y = synth_patches(B, N, seed=2).cuda()
    d = torch.empty(B, N, device="cuda"); i = torch.empty(B, N, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.pf_nn1(x.data_ptr(), y.data_ptr(), B, N, N, d.data_ptr(), i.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lib.pf_nn1(x.data_ptr(), y.data_ptr(), B, N, N, d.data_ptr(), i.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    print(f"nn1 {B} x {N}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us", flush=True)
