#!/usr/bin/env python3
"""Wall time of the .xyz CLI (puflow_amd.upsample.upsampling) per cloud: a directory of equal-size clouds, one file at a
time (the reference's loop) against batches of files.  Includes reading, the GPU pipeline and writing.
  python tools/time_cli.py [n_points] [n_files]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from puflow_amd import upsample as U
from puflow_amd.weights import synth_patches, synth_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sd = synth_state_dict(2021)
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "in"); os.makedirs(src)
    for k in range(F):
        np.savetxt(os.path.join(src, f"cloud{k:03d}.xyz"), synth_patches(1, N, seed=100 + k)[0].numpy(), fmt="%.6f")
    paths = sorted(os.path.join(src, f) for f in os.listdir(src))
    for cb in (1, 1, 8, 16, 32):
        dst = os.path.join(tmp, f"out{cb}"); os.makedirs(dst, exist_ok=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        U.upsampling(paths, dst, None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd, cloud_batch=cb)
        dt = time.perf_counter() - t0
        print(f"cloud_batch {cb:2d}: {F} clouds of {N} -> {4 * N} points in {dt:6.3f} s = {dt / F * 1e3:7.2f} ms per cloud", flush=True)
