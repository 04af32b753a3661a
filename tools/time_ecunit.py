#!/usr/bin/env python3
"""GPU box: one EdgeConv unit of the training step, forward (and backward) as the per-layer launches vs the persistent
grid-barrier launch (csrc/train_fused.hip ec_fwdp_kernel), HIP-event times at the training shape 32 x 256 points, K = 16.
  python tools/time_ecunit.py [B] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops, train_ops
from puflow_amd.interpflow import _EdgeConvParams
from puflow_amd.weights import synth_patches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
xyz = synth_patches(B, N, seed=3).cuda()
idx, _ = ops.knn_idx32(xyz, xyz, 16)
csr = train_ops.knn_csr(idx)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


UNITS = ((3, 32, 8), (32, 64, 16), (64, 128, 32), (128, 128, 32))
if os.environ.get("PF_ECUNIT_ONLY"):
    UNITS = (UNITS[-1],)
for cin, odim, g in UNITS:
    torch.manual_seed(g)
    p = _EdgeConvParams(cin, odim, g).cuda().train()
    x = xyz if cin == 3 else torch.randn(B, N, cin, device="cuda")
    for persistent in (False, True):
        with torch.no_grad():
            tf = timed(lambda: train_ops.edgeconv_train_fused(p, x, idx, True, csr, persistent))
        xx = x.clone().requires_grad_(True)

        def fb():
            out = train_ops.edgeconv_train_fused(p, xx, idx, True, csr, persistent)
            out.sum().backward()
        tfb = timed(fb)
        print(f"unit C={cin:3d} g={g:2d} odim={odim:3d}  persistent={int(persistent)}  forward {tf:7.1f} us   forward+backward {tfb:7.1f} us", flush=True)
train_ops.check_persist_status()
