#!/usr/bin/env python3
"""GPU box: does the interpolation kernel (independent of the feature extractor until its last step) overlap with the EdgeConv
chain when both are enqueued on different streams?  Sequential vs two-stream time of [EdgeConv units 2-5 + P|Q GEMMs] and
[interp]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
from puflow_amd.interpflow import PointInterpFlow, FEAT_CHANNELS
from puflow_amd.weights import synth_patches, synth_state_dict
B, N = 32, 2048
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.cuda().eval()
xyz = synth_patches(B, N, seed=2021).cuda()
eng = net._engine() if hasattr(net, "_engine") else None
T = B * N
with torch.no_grad():
    idx16 = eng.knn(xyz)
    z = torch.randn(B, N, 3, device="cuda")
    pq = torch.randn(T, 512, device="cuda") * 0.1
    hs = [torch.empty((T, FEAT_CHANNELS[u + 1]), device="cuda") for u in range(6)]
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def chain(stream):
        with torch.cuda.stream(stream):
            s = stream.cuda_stream
            for u in range(2, 6):
                eng._edgeconv(u, pq.data_ptr(), idx16, hs[u], B, N, s)
                if u + 1 < 6:
                    _lib.check(eng.lib.pf_pq_gemm(u, hs[u].data_ptr(), eng.base, eng.post[u], pq.data_ptr(), T, s))

    def interp(stream):
        with torch.cuda.stream(stream):
            return eng.interp(xyz, z, idx16, 4)

    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_chain = timeit(lambda: chain(sA))
    t_int = timeit(lambda: interp(sA))
    t_seq = timeit(lambda: (chain(sA), interp(sA)))
    t_par = timeit(lambda: (chain(sA), interp(sB)))
    print(f"chain {t_chain:.3f} ms  interp {t_int:.3f} ms  same stream {t_seq:.3f} ms  two streams {t_par:.3f} ms")
