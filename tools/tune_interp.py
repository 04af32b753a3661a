#!/usr/bin/env python3
"""GPU box: interp_kernel launch shapes (P column tiles per wave, NW waves per workgroup) as side-by-side libraries:
    python tools/tune_interp.py build          (build container: libpuflow_hip_ip<P>_<NW>.so)
    python tools/tune_interp.py [B] [N]        (GPU box: ms per launch, output bit-equality with the shipped shape)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puflow_amd import build as Bd

SHAPES = [(2, 8), (2, 6), (1, 8), (1, 10)]
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for p, nw in SHAPES:
        Bd.build(defines=[f"PF_INTERP_P={p}", f"PF_INTERP_NW={nw}"], tag=f"ip{p}_{nw}", only=("interp.hip",), verbose=False)
    sys.exit(0)

import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
e = net._engine(4)
idx16 = e.knn(xyz)
z = torch.randn(B, N, 3, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def load(tag):
    l = ctypes.CDLL(_lib.LIB_PATH.replace(".so", f"_{tag}.so") if tag else _lib.LIB_PATH)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    return l


libs = [("shipped", load(""))] + [(f"({p},{nw})", load(f"ip{p}_{nw}")) for p, nw in SHAPES]
outs, times = {}, {k: [] for k, _ in libs}
for rnd in range(6):
    for name, l in libs:
        u = torch.empty((B, N * 4, 3), device="cuda")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            rc = l.pf_interp(xyz.data_ptr(), z.data_ptr(), idx16.data_ptr(), e.base, e.interp_off, u.data_ptr(), B, N, 4, s)
        b.record(); torch.cuda.synchronize()
        assert rc == 0
        if rnd:
            times[name].append(a.elapsed_time(b) / 5)
        outs[name] = u
ref = outs["shipped"]
for name, t in times.items():
    t = sorted(t)
    print(f"{name:16s} min {t[0] * 1e3:8.1f} us  med {t[len(t) // 2] * 1e3:8.1f} us   bit-equal to shipped: {bool(torch.equal(outs[name], ref))}", flush=True)
