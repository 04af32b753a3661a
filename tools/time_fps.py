#!/usr/bin/env python3
"""Times pf_fps on the CLI's merge shape (99 840 -> 20 024 points; 1, 8 and 32 clouds at once).
  python tools/time_fps.py [n_points] [n_sample] [n_clouds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 99840
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20024
for B in [int(b) for b in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 8, 32]:
    g = torch.Generator().manual_seed(7)
    pc = torch.rand(B, N, 3, generator=g).cuda()
    idx = ops.furthest_point_sample(pc, M)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 3
    for _ in range(iters):
        idx = ops.furthest_point_sample(pc, M)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"B={B} N={N} -> {M}: {dt * 1e3:8.2f} ms  ({dt / (M - 1) * 1e6:.3f} us / step)", flush=True)
