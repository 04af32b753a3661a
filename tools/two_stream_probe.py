#!/usr/bin/env python3
"""GPU box: does replaying two captured forwards on two streams (consecutive steps pipelined) raise the throughput over
one stream?  32 x 2048 patches per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict
B, N = 32, 2048
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
with torch.no_grad():
    runs, streams = [], []
    for k in range(2):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            runs.append(net.graphed(B, N, 4))
        streams.append(s)
    torch.cuda.synchronize()

    def bench(nstreams, steps=200):
        for w in range(10):
            with torch.cuda.stream(streams[w % nstreams]):
                runs[w % nstreams](xyz)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % nstreams]):
                runs[i % nstreams](xyz)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for n in (1, 2, 1, 2):
        ms = bench(n)
        print(f"{n} stream(s): {ms:.4f} ms / step = {B / ms * 1e3:.0f} patches/s", flush=True)
