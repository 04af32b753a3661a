#!/usr/bin/env python3
"""GPU box: edgeconv1n_kernel (units 0 / 1 in the product arithmetic) - time per launch of the launch shapes (points per wave P,
waves per workgroup NW) at B x N points (env B, N; default 4 x 2048: the per-GPU share of one 32-patch batch on 8 GPUs).
Needs the -DPF_TUNING_VARIANTS build:
    python -c "from puflow_amd import build as b; b.build(defines=['PF_TUNING_VARIANTS'], tag='abl', only=('edgeconv.hip',))"
    B=4 python tools/tune_ec1n.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

B, N = int(os.environ.get("B", 4)), int(os.environ.get("N", 2048))
T = B * N
xyz = synth_patches(B, N, seed=2021).cuda()
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
e = net._engine(4)
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
idx16 = e.knn(xyz)
pq = torch.empty((T, 512), device="cuda")
h0 = torch.empty((T, 32), device="cuda")
e._edgeconv(0, xyz.data_ptr(), idx16, h0, B, N, s)
st = torch.empty((6, T, 8), device="cuda"); cp = torch.empty((6, T, 64), device="cuda")
_lib.check(lib.pf_post(0, h0.data_ptr(), e.base, e.post[0], None, st[0].data_ptr(), cp[0].data_ptr(), pq.data_ptr(), T, s))
torch.cuda.synchronize()
abl = ctypes.CDLL(_lib.LIB_PATH.replace(".so", "_abl.so"))
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(abl, name); fn.restype, fn.argtypes = res, args
shapes = {0: "(P1,NW8)", 1: "(P2,NW8) shipped", 2: "(P1,NW16)", 3: "(P2,NW4)"}
for unit, cfg, src, od in ((0, 8, xyz, 32), (1, 9, pq, 64)):
    ref = torch.empty((T, od), device="cuda")
    assert abl.pf_edgeconv_tuned(cfg, 1, src.data_ptr(), None, idx16.data_ptr(), e._p(e.ec1n_w[unit]), ref.data_ptr(), B, N, s) == 0
    times = {v: [] for v in shapes}
    for rnd in range(9):
        for v in shapes:
            out = torch.empty_like(ref)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                rc = abl.pf_edgeconv_tuned(cfg, v, src.data_ptr(), None, idx16.data_ptr(), e._p(e.ec1n_w[unit]), out.data_ptr(), B, N, s)
            b.record(); torch.cuda.synchronize()
            assert rc == 0, (unit, v, rc)
            if rnd == 0:
                assert torch.equal(out, ref), (unit, v, "launch shape changed the bits")
            else:
                times[v].append(a.elapsed_time(b) / 5)
    for v, t in times.items():
        t = sorted(t)
        print(f"unit {unit}  {shapes[v]:18s} min {t[0] * 1e3:6.1f} us  med {t[len(t) // 2] * 1e3:6.1f} us", flush=True)
