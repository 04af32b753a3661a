#!/usr/bin/env python3
"""GPU box: wall time of every pf_fps call inside the CLI's one-file loop (cloud_batch = 1), and of the stages around it."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puflow_amd import upsample as U, ops, patch as PT
from puflow_amd.weights import synth_patches, synth_state_dict
sd = synth_state_dict(2021)
rec = []
LAST = {"rounds": None}
import ctypes
from puflow_amd import _lib
_orig_check = ops._check_fps_abort
def _check(lib, mind, B, N):
    stride, word = ctypes.c_longlong(0), ctypes.c_longlong(0)
    if lib.pf_fps_scratch_layout(N, ctypes.byref(stride), ctypes.byref(word)):
        w = mind.view(-1)[: (B * N) // 2 * 2].view(torch.int64)
        LAST["rounds"] = [int(w[b * stride.value + word.value + 1].item()) for b in range(B)]
    else:
        LAST["rounds"] = None
    return _orig_check(lib, mind, B, N)
ops._check_fps_abort = _check
orig = ops.furthest_point_sample
def timed(xyz, npoint, group=0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = orig(xyz, npoint, group=group)
    b.record()
    torch.cuda.synchronize()
    rounds = LAST["rounds"]
    rec.append((tuple(xyz.shape), npoint, group, round((time.perf_counter() - t0) * 1e3, 2), round(a.elapsed_time(b), 2), rounds))
    return r
ops.furthest_point_sample = timed
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "in"); os.makedirs(src)
    for k in range(12):
        np.savetxt(os.path.join(src, f"cloud{k:03d}.xyz"), synth_patches(1, 5000, seed=100 + k)[0].numpy(), fmt="%.6f")
    paths = sorted(os.path.join(src, f) for f in os.listdir(src))
    for cb in (1, 4):
        rec.clear(); os.makedirs(os.path.join(tmp, f"o{cb}"))
        t0 = time.perf_counter()
        U.upsampling(paths, os.path.join(tmp, f"o{cb}"), None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd, cloud_batch=cb)
        print(f"cloud_batch {cb}: {(time.perf_counter() - t0) / len(paths) * 1e3:.1f} ms per cloud")
        for r in rec: print("   ", r)
