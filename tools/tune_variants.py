#!/usr/bin/env python3
"""A/B of compile-time tuning variants (P = column tiles per wave, NW = waves per workgroup) of the
post / flow / interp kernels: builds libpuflow_hip_<tag>.so per variant (HERE, before gpurun) and
times them interleaved in ONE process on the GPU (guide rule 24).
  python tools/tune_variants.py build      # in the build container
  python tools/tune_variants.py run        # on the GPU box
"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {
    "a": ["PF_INTERP_P=2", "PF_INTERP_NW=8", "PF_MM2_DEPTH=4"],
    "b": ["PF_INTERP_P=1", "PF_INTERP_NW=8", "PF_MM2_DEPTH=4"],
    "c": ["PF_INTERP_P=2", "PF_INTERP_NW=8", "PF_MM2_DEPTH=3"],
}
if sys.argv[1] == "build":
    from puflow_amd import build
    for tag, d in VARIANTS.items():
        print(build.build(defines=d, tag=tag, verbose=False))
    sys.exit(0)

import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict
libs = {"base": _lib.load()}
for tag in VARIANTS:
    l = ctypes.CDLL(_lib.LIB_PATH.replace(".so", f"_{tag}.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    libs[tag] = l
B, N = 32, 2048
sd = synth_state_dict(2021)
net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
e = net._engine(4)
ref = None
acc = {}
for rnd in range(5):
    for tag, l in libs.items():
        e.lib = l
        pr = e.profile_stages(xyz, iters=3)
        x, _ = net(xyz, 4)
        if ref is None: ref = x.clone()
        assert torch.equal(x, ref), tag
        if rnd:
            for k, v in pr.items(): acc.setdefault(k, {}).setdefault(tag, []).append(v)
for k in acc:
    if k.startswith("edgeconv"): continue
    print(f"{k:12s}", {t: f"{min(v):.3f}" for t, v in acc[k].items()}, flush=True)
