#!/usr/bin/env python3
"""GPU box: board power and shader clock while ONE kernel of the eval step runs in a loop (rocm-smi sampled every ~0.2 s from a
side thread) - the counter-level evidence behind "edgeconv4_kernel is limited by energy per point as much as by issue slots"
(DESIGN section 4): the clock the chip holds under each kernel against the clock it holds idle / under an HBM-bound kernel.
python tools/power_trace.py [seconds per phase]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B, N = 32, 2048
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
e = net._engine(4)
lib = _lib.load()
T = B * N
s = torch.cuda.current_stream().cuda_stream
idx16 = e.knn(xyz)
st = net.forward_stages(xyz, 4)
pq = torch.randn(T, 512, device="cuda") * 0.5
h = torch.empty(T, 128, device="cuda")
z = st["z"].contiguous()
u = torch.empty(B, N * 4, 3, device="cuda")


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            pw = [float(v) for k, v in card.items() if "ower" in k and "W" in k and v not in ("N/A", "")]
            ck = [v for k, v in card.items() if "sclk" in k.lower()]
            out.append((time.time(), pw[0] if pw else None, ck[0] if ck else None))
        except Exception as ex:                      # keep sampling; report what was seen
            out.append((time.time(), None, repr(ex)[:60]))
        time.sleep(0.15)


def phase(name, fn):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < SEC:
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); n += 50
    el = time.time() - t0
    stop.set(); th.join()
    out = out[2:] or out                              # the first samples predate the ramp
    pw = [p for _, p, _ in out if p is not None]
    ck = [c for _, _, c in out if c]
    print(f"{name:34s} {el / n * 1e6:8.1f} us/launch   power {sum(pw) / max(len(pw), 1):6.0f} W (max {max(pw) if pw else 0:.0f})   sclk samples {sorted(set(ck))[:6]}  ({len(out)} samples)", flush=True)


phase("idle (sleep)", lambda: time.sleep(0.002))
phase("edgeconv4_kernel (unit 3)", lambda: _lib.check(lib.pf_edgeconv(7, pq.data_ptr(), None, idx16.data_ptr(), e._p(e.ec4_w[3]), h.data_ptr(), B, N, s)))
phase("pq_gemm_kernel (HBM write bound)", lambda: _lib.check(lib.pf_pq_gemm(3, h.data_ptr(), e.base, e.post[3], pq.data_ptr(), T, s)))
phase("interp_kernel", lambda: _lib.check(lib.pf_interp(xyz.data_ptr(), z.data_ptr(), idx16.data_ptr(), e.base, e.interp_off, u.data_ptr(), B, N, 4, s)))
phase("knn4_kernel (VALU bound)", lambda: e.knn(xyz))
