#!/usr/bin/env python3
"""GPU box: where knn4_kernel's time goes at 32 x 2048, K = 16 - the shipped kernel and timing-only ablation builds
(-DPF_KNN_ABL=1 no merge, =2 sweep A only).  Build the variants first (here or on the box):  python tools/tune_knn.py build"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puflow_amd import build as Bd

VARIANTS = {"knn1": ["PF_KNN_ABL=1"], "knn2": ["PF_KNN_ABL=2"]}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for tag, d in VARIANTS.items():
        Bd.build(defines=d, tag=tag, only=("knn.hip",), verbose=False)
    sys.exit(0)

import torch
from puflow_amd import _lib
from puflow_amd.weights import synth_patches

B, N, K = int(os.environ.get("B", 32)), int(os.environ.get("N", 2048)), 16
xyz = synth_patches(B, N, seed=2021).cuda()
idx = torch.empty((B, N, K), dtype=torch.int32, device="cuda")


def load(tag):
    l = ctypes.CDLL(_lib.LIB_PATH.replace(".so", f"_{tag}.so") if tag else _lib.LIB_PATH)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    return l


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


s = torch.cuda.current_stream().cuda_stream
for tag, what in (("", "shipped"), ("knn1", "no merge"), ("knn2", "sweep A + threshold only")):
    l = load(tag)
    t = timed(lambda: l.pf_knn(xyz.data_ptr(), xyz.data_ptr(), B, N, N, K, idx.data_ptr(), None, s))
    print(f"{what:28s} {t * 1e3:7.1f} us", flush=True)
