#!/usr/bin/env python3
"""GPU box: the EMD auction alone at the training shape (32 x 1024 points, eps 0.005, 50 iterations): a far-off prediction
(most points stay unassigned for all iterations: the expensive case early in training) and a near one.
  python tools/time_emd.py [groups ...]     workgroups per sample to sweep (default 0 = chosen from the device; 1 = one workgroup per sample)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import loss as L
from puflow_amd.weights import synth_patches

B, n = 32, 1024
gt = ((synth_patches(B, n, seed=1) + 1) / 2).cuda()
GROUPS = [int(v) for v in sys.argv[1:]] or [0]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


torch.manual_seed(0)
cases = (("far (random cloud)", torch.rand(B, n, 3, device="cuda")),
         ("near (target + 1 % noise)", (gt + 0.01 * torch.randn(B, n, 3, device="cuda")).clamp(0, 1)),
         ("same cloud, shuffled", gt[:, torch.randperm(n)].contiguous()))
for g in GROUPS:
    emd = L.EarthMoverDistance(eps=0.005, iters=50, groups=g)
    for name, pred in cases:
        with torch.no_grad():
            t = timed(lambda: emd(pred, gt))
        print(f"groups {g:2d}  {name:28s} {t * 1e3:8.1f} us", flush=True)
L.check_emd_status()
