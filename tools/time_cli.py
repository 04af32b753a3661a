#!/usr/bin/env python3
"""Wall time of the .xyz CLI (puflow_amd.upsample.upsampling) per cloud over a directory of equal-size clouds: one file at a
time (the reference's loop) against batches of files.  Every configuration runs in a FRESH process (as the CLI does), timed
from the call of upsampling() - weight plan, first allocations, reading, the GPU pipeline and writing included; library import
and GPU context creation (the same for all configurations) excluded.
  python tools/time_cli.py [n_points] [n_files]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    src, dst, cb = sys.argv[2], sys.argv[3], int(sys.argv[4])
    import torch
    from puflow_amd import upsample as U
    from puflow_amd.weights import synth_state_dict
    sd = synth_state_dict(2021)
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()           # GPU context
    paths = sorted(os.path.join(src, f) for f in os.listdir(src))
    t0 = time.perf_counter()
    U.upsampling(paths, dst, None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd, cloud_batch=cb)
    print("ELAPSED %.6f" % (time.perf_counter() - t0), flush=True)
    sys.exit(0)

import numpy as np
from puflow_amd.weights import synth_patches

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "in"); os.makedirs(src)
    for k in range(F):
        np.savetxt(os.path.join(src, f"cloud{k:03d}.xyz"), synth_patches(1, N, seed=100 + k)[0].numpy(), fmt="%.6f")
    for cb in (1, 8, 16, 32):
        dst = os.path.join(tmp, f"out{cb}"); os.makedirs(dst, exist_ok=True)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", src, dst, str(cb)], capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("ELAPSED")]
        if out.returncode != 0 or not line:
            print(f"cloud_batch {cb:2d}: FAILED\n{out.stderr[-800:]}")
            continue
        dt = float(line[-1].split()[1])
        print(f"cloud_batch {cb:2d}: {F} clouds of {N} -> {4 * N} points in {dt:6.3f} s = {dt / F * 1e3:7.2f} ms per cloud (fresh process)", flush=True)
