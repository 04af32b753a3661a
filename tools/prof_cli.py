#!/usr/bin/env python3
"""cProfile of the CLI's upsampling() over a directory of clouds (where the wall time per cloud goes on the host side).
  python tools/prof_cli.py [n_points] [n_files] [cloud_batch]"""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from puflow_amd import upsample as U
from puflow_amd.weights import synth_patches, synth_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
CB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sd = synth_state_dict(2021)
with tempfile.TemporaryDirectory() as tmp:
    src = os.path.join(tmp, "in"); os.makedirs(src)
    for k in range(F):
        np.savetxt(os.path.join(src, f"cloud{k:03d}.xyz"), synth_patches(1, N, seed=100 + k)[0].numpy(), fmt="%.6f")
    paths = sorted(os.path.join(src, f) for f in os.listdir(src))
    run = lambda tag: U.upsampling(paths, os.path.join(tmp, tag), None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021,
                                   state_dict=sd, cloud_batch=CB)
    os.makedirs(os.path.join(tmp, "w")); os.makedirs(os.path.join(tmp, "p"))
    run("w")                                       # warm-up (library load, first launches)
    pr = cProfile.Profile(); pr.enable(); run("p"); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
