#!/usr/bin/env python3
"""GPU box: edgeconv4_kernel (the product arithmetic, ec_mode "f16n") on unit 3 at 32 x 2048: time per launch (HIP events,
interleaved rounds, one process) and deviation from the bit-exact f32 MFMA kernel.  The default library carries only the
shipped launch shape; the alternatives and the timing-only ablations need a -DPF_TUNING_VARIANTS build:
    python -c "from puflow_amd import build as b; b.build(defines=['PF_TUNING_VARIANTS'], tag='abl', only=('edgeconv.hip',))"
    python tools/tune_ec4.py abl"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

B, N = int(os.environ.get("B", 32)), int(os.environ.get("N", 2048))
sd = synth_state_dict(2021)
xyz = synth_patches(B, N, seed=2021).cuda()
lib = _lib.load()
T = B * N
s = torch.cuda.current_stream().cuda_stream


def unit_inputs(mode, upto=3):
    """P|Q table that feeds unit `upto` and the unit's output under `mode`."""
    net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.cuda().eval()
    net.ec_mode = mode
    e = net._engine(4)
    idx16 = e.knn(xyz)
    cp = torch.empty((6, T, 64), device="cuda"); st = torch.empty((6, T, 8), device="cuda")
    pq = torch.empty((T, 512), device="cuda")
    for u in range(upto + 1):
        h = torch.empty((T, [32, 64, 128, 128, 128, 128][u]), device="cuda")
        src = xyz.data_ptr() if u == 0 else pq.data_ptr()
        if u == upto:
            keep = pq.clone()
            src = keep.data_ptr()
        e._edgeconv(u, src, idx16, h, B, N, s)
        _lib.check(lib.pf_post(u, h.data_ptr(), e.base, e.post[u], None, st[u].data_ptr(), cp[u].data_ptr(),
                               pq.data_ptr() if u < 5 else None, T, s))
    torch.cuda.synchronize()
    return e, idx16, keep, h


e32, idx16, pq32, h32 = unit_inputs("f32")
en, _, pqn, hn = unit_inputs("f16n")
print(f"unit 3 |h|max {float(h32.abs().max()):.2f}   f16n vs f32 max|d| {float((hn - h32).abs().max()):.3e}", flush=True)

cands = [("f16n v0 (P1,NW16)", 7, 0, en, pqn, "ec4_w", lib)]
# side-by-side tuning builds: python tools/tune_ec4.py <tag> ...  loads libpuflow_hip_<tag>.so (puflow_amd.build.build(defines, tag))
import ctypes
for tag in sys.argv[1:]:
    l = ctypes.CDLL(_lib.LIB_PATH.replace(".so", f"_{tag}.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name); fn.restype, fn.argtypes = res, args
    if tag == "abl":       # -DPF_TUNING_VARIANTS build: timing-only ablations of the (P1, NW16) shape
        cands += [(f"[abl] {what}", 7, v, en, pqn, "ec4_w", l) for v, what in
                  ((0, "full"), (1, "shape (P2,NW8)"), (2, "shape (P1,NW8)"), (3, "shape (P2,NW4)"), (8, "no gathers"), (9, "no MFMAs"), (10, "no LDS weight reads"), (11, "no gathers, no LDS reads"),
                   (12, "no gathers, no MFMAs"), (13, "no MFMAs, no LDS reads"), (14, "no gathers, LDS reads, growth epilogues"))]
        continue
    cands += [(f"[{tag}] f16n v0 (P1,NW16)", 7, 0, en, pqn, "ec4_w", l), (f"[{tag}] f16n v1 (P2,NW8)", 7, 1, en, pqn, "ec4_w", l)]
times = {c[0]: [] for c in cands}
for rnd in range(8):
    for name, cfg, v, e, pq, wname, lib in cands:
        out = torch.empty_like(h32)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            rc = lib.pf_edgeconv_tuned(cfg, v, pq.data_ptr(), None, idx16.data_ptr(), e._p(getattr(e, wname)[3]), out.data_ptr(), B, N, s)
        b.record(); torch.cuda.synchronize()
        assert rc == 0, (name, rc)
        if rnd > 0:
            times[name].append(a.elapsed_time(b) / 5)
        if rnd == 0:
            if "[abl]" not in name or "full" in name:
                print(f"  {name}: max|d| vs f32 {float((out - h32).abs().max()):.3e}", flush=True)
for name, t in times.items():
    t = sorted(t)
    print(f"{name:36s} min {t[0]:.4f} ms  med {t[len(t) // 2]:.4f} ms", flush=True)
