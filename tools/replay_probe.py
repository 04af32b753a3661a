#!/usr/bin/env python3
"""GPU box: is a small-batch graph replay bound by the GPU or by the host?  Host time of the pieces of PointInterpFlow.graphed()'s
run() (plan signature check, input copy, hipGraphLaunch) against the GPU time per step (HIP events around 200 replays) and the
wall time per step.   python tools/replay_probe.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
run = net.graphed(B, N, 4)
for _ in range(20): run(xyz)
torch.cuda.synchronize()
n = 200
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); a.record()
for _ in range(n): run(xyz)
b.record(); th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"run():         host enqueue {th / n * 1e3:.4f} ms/step   wall {tw / n * 1e3:.4f} ms/step   GPU (events) {a.elapsed_time(b) / n:.4f} ms/step")
g = run.graph
torch.cuda.synchronize(); t0 = time.perf_counter(); a.record()
for _ in range(n): g.replay()
b.record(); th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"graph.replay:  host enqueue {th / n * 1e3:.4f} ms/step   wall {tw / n * 1e3:.4f} ms/step   GPU (events) {a.elapsed_time(b) / n:.4f} ms/step")
t0 = time.perf_counter()
for _ in range(n): net._engine(4)
print(f"plan signature check: {(time.perf_counter() - t0) / n * 1e3:.4f} ms")
# back-to-back replays keep the GPU queue full: per-step GPU time = the kernels + inter-kernel gaps; a single replay in
# isolation (sync before and after) adds the launch latency
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"one replay, synchronised: {min(ts) * 1e3:.4f} ms (min of 20)")
