#!/usr/bin/env python3
"""Timing of the continuous (CNF) x4 forward on the GPU (BASELINE configs[4]): ms per forward, function evaluations,
time per right-hand side.   python tools/time_cnf.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.cnf import PointInterpFlow
from puflow_amd.weights import CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES, synth_cnf_state_dict, synth_patches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
# the bench workload (bench.py --mode cnf): the trained checkpoint's end times, ODE nets scaled until dopri5 works like on it
net = PointInterpFlow(3); net.load_state_dict(synth_cnf_state_dict(2021, dynamics=CNF_PU1K_DYNAMICS, end_times=CNF_PU1K_END_TIMES)); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
torch.manual_seed(0)
noise = [torch.randn(B, N, 3, device="cuda") for _ in range(6)]
for it in range(int(os.environ.get("PF_TIME_CNF_ITERS", "3"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, logp = net(xyz, 4, noise=noise)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"forward {B} x {N} -> x4: {el * 1e3:8.2f} ms  ({B / el:8.1f} patches/s)  {net.last_stats}", flush=True)
if os.environ.get("PF_TIME_CNF_NO_RHS"):        # tools/trace_sequence.sh: the trace ends with a forward
    raise SystemExit(0)
eng = net._engine(4)
T = B * N
ctx = torch.randn(T, 288, device="cuda"); e = noise[0].reshape(T, 3)
for rows, R in ((T, 1), (4 * T, 4)):
    y = torch.randn(rows, 4, device="cuda"); out = torch.empty_like(y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng._rhs(2, y, y, [], 0.0, 0.1, 1.0, ctx, e, out, None, rows, R)
    a.record()
    for _ in range(20):
        eng._rhs(2, y, y, [], 0.0, 0.1, 1.0, ctx, e, out, None, rows, R)
    b.record(); torch.cuda.synchronize()
    print(f"rhs rows={rows}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us", flush=True)
