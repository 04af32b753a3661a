#!/usr/bin/env python3
"""In-process A/B timing of the EdgeConv launch variants (guide rule 24: interleaved rounds, one process)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow, _EC_CFG
from puflow_amd.packing import FEAT_CHANNELS
from puflow_amd.weights import synth_patches, synth_state_dict

B, N = 32, 2048
sd = synth_state_dict(2021)
net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
e = net._engine(4)
lib = _lib.load()
idx16 = e.knn(xyz)
T = B * N
s = torch.cuda.current_stream().cuda_stream
# build inputs for every unit with the default path
pqs, hs = {}, {}
cp = torch.empty((6, T, 64), device="cuda"); st = torch.empty((6, T, 8), device="cuda")
pq = torch.empty((T, 512), device="cuda")
for u in range(6):
    h = torch.empty((T, FEAT_CHANNELS[u + 1]), device="cuda")
    src = xyz if u == 0 else pq.clone()
    pqs[u] = src
    tab = e._p(e.ec_tab0) if u == 0 else None
    _lib.check(lib.pf_edgeconv(_EC_CFG[u], src.data_ptr(), tab, idx16.data_ptr(), e._p(e.ec_w[u]), h.data_ptr(), B, N, s))
    hs[u] = h
    _lib.check(lib.pf_post(u, h.data_ptr(), e.base, e.post[u], None, st[u].data_ptr(), cp[u].data_ptr(),
                           pq.data_ptr() if u < 5 else None, T, s))
torch.cuda.synchronize()
res = {}
for u in (0, 1, 3):
    tab = e._p(e.ec_tab0) if u == 0 else None
    times = {v: [] for v in range(4)}
    for rnd in range(6):
        for v in range(4):
            out = torch.empty_like(hs[u])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.pf_edgeconv_tuned(_EC_CFG[u], v, pqs[u].data_ptr(), tab, idx16.data_ptr(), e._p(e.ec_w[u]),
                                       out.data_ptr(), B, N, s)
            b.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            assert torch.equal(out, hs[u]), f"variant {v} unit {u} differs"
            if rnd > 0:
                times[v].append(a.elapsed_time(b))
    res[u] = {v: (min(t), sorted(t)[len(t) // 2]) for v, t in times.items()}
    print("unit", u, {v: f"min {a:.3f} med {b:.3f} ms" for v, (a, b) in res[u].items()}, flush=True)

# ---- split-bf16 variants of unit 3 (cfg 3): speed + deviation from the exact-f32 kernel
u = 3
VARS = (0, 1, 2, 8, 9) if '--ablate' in sys.argv else (0, 1, 2)   # 8/9 need a -DPF_TUNING_VARIANTS build
times = {v: [] for v in VARS}
dev = {}
for rnd in range(6):
    for v in VARS:
        out = torch.empty_like(hs[u])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.pf_edgeconv_tuned(3, v, pqs[u].data_ptr(), None, idx16.data_ptr(), e._p(e.ec3_w[u]), out.data_ptr(), B, N, s)
        b.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        dev[v] = (float((out - hs[u]).abs().max()), float(hs[u].abs().max()))
        if rnd > 0:
            times[v].append(a.elapsed_time(b))
print("unit 3 bf16x3", {v: f"min {min(t):.3f} ms  max|d| {dev[v][0]:.2e} (|h|max {dev[v][1]:.2f})" for v, t in times.items()}, flush=True)

# ---- split-fp16 variants of unit 3 (cfg 4); 8..11 = ablations (need a -DPF_TUNING_VARIANTS build)
VARS = (2, 8, 9, 10, 11, 12, 13, 14) if '--ablate' in sys.argv else (0, 1, 2, 3, 4)
times = {v: [] for v in VARS}
dev = {}
for rnd in range(6):
    for v in VARS:
        out = torch.empty_like(hs[u])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.pf_edgeconv_tuned(4, v, pqs[u].data_ptr(), None, idx16.data_ptr(), e._p(e.ec2h_w[u]), out.data_ptr(), B, N, s)
        b.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        dev[v] = (float((out - hs[u]).abs().max()), float(hs[u].abs().max()))
        if rnd > 0:
            times[v].append(a.elapsed_time(b))
print("unit 3 f16x2", {v: f"min {min(t):.3f} ms  max|d| {dev[v][0]:.2e} (|h|max {dev[v][1]:.2f})" for v, t in times.items()}, flush=True)

# ---- split-fp16 kernels of the narrow units 0 / 1 (cfg 5 / 6)
for u in (0, 1):
    VARS = (0, 1, 2, 3, 4)
    times = {v: [] for v in VARS}
    dev = {}
    for rnd in range(6):
        for v in VARS:
            out = torch.empty_like(hs[u])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.pf_edgeconv_tuned(5 + u, v, pqs[u].data_ptr(), None, idx16.data_ptr(), e._p(e.ec1h_w[u]), out.data_ptr(), B, N, s)
            b.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            dev[v] = (float((out - hs[u]).abs().max()), float(hs[u].abs().max()))
            if rnd > 0:
                times[v].append(a.elapsed_time(b))
    print(f"unit {u} f16x2", {v: f"min {min(t):.3f} ms  max|d| {dev[v][0]:.2e} (|h|max {dev[v][1]:.2f})" for v, t in times.items()}, flush=True)
