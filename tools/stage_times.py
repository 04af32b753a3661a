#!/usr/bin/env python3
"""GPU box: per-stage milliseconds of the eval forward at B x N (HIP events on the launch stream), eager and graph-replayed
step time.  python tools/stage_times.py [B] [N] [fuse]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
FUSE = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # pf_edgeconv_pq: -1 by size, 0 never, 1 always
SPLIT = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # interpolation weights as a parallel branch: 0 off (default), 1 on
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
e = net._engine(4)
e.fuse_pq = FUSE
e.split_interp = SPLIT
pr = e.profile_stages(xyz, iters=8)
tot = sum(pr.values())
for k, v in pr.items():
    print(f"{k:14s} {v:8.4f} ms  {100 * v / tot:5.1f} %")
print(f"{'sum':14s} {tot:8.4f} ms")
run = net.graphed(B, N, 4)
for _ in range(20): run(xyz)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): run(xyz)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 100
print(f"graph replay   {el * 1e3:8.4f} ms / step  = {B / el:9.0f} patches/s")
