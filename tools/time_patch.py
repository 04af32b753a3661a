#!/usr/bin/env python3
"""Stage timing of the .xyz upsampling pipeline (PatchHelper.upsample, modules/utils/patch.py:35-80) on the GPU:
which part of the CLI's per-cloud time is the network and which the patch operators.
  python tools/time_patch.py [n_points] [n_clouds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.patch import PatchHelper
from puflow_amd.weights import synth_patches, synth_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda"
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.to(dev).eval()
pc = synth_patches(B, N, seed=5).to(dev)


def timed(name, fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / iters * 1e3:9.3f} ms", flush=True)
    return r


ph = PatchHelper(256, 4.0)
n_patch = int(N / 256 * 4)
with torch.no_grad():
    timed("whole PatchHelper.upsample", lambda: ph.upsample(net, pc, npoint=N * 4, upratio=4))
    pcn, _, _ = PatchHelper.normalize_pc(pc)
    seeds = timed(f"fps seeds {N}->{n_patch}", lambda: ops.furthest_point_sample(pcn, n_patch))
    patches = timed("knn K=256 patches", lambda: PatchHelper.extract_knn_patch(pcn, ph.knn, 256, 4.0))
    flat = patches.reshape(B * n_patch, 256, 3)
    pn, _, _ = PatchHelper.normalize_pc(flat)
    pn = pn.contiguous()
    timed(f"network {B * n_patch} x 256 -> 1024", lambda: net.sample(pn, upratio=4))
    pred = PatchHelper.upsampling_patches(net, patches, 4)
    M = pred.shape[1] * pred.shape[2]
    timed(f"fps merge {M}->{N * 4 + 24}... ", lambda: PatchHelper.merge_patches(pred, N * 4))
    cloud = pred.reshape(B, M, 3)[:, :N * 4].contiguous()
    timed("nearest distance (outliers)", lambda: ops.nearest_distance(cloud, pc))
