#!/usr/bin/env python3
"""GPU box: the three point GEMMs of every EdgeConv unit of the training step (32 x 256 points), pf_gemm_ex arith 0
(gemm2_kernel: conflict-free LDS images, round 5) against arith 1 (gemm_kernel, round 1): HIP events, interleaved rounds, one
process, bit equality.   python tools/time_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import train_ops as T

Tn = 8192
units = [(3, 32, 8), (32, 64, 16), (64, 128, 32), (128, 128, 32)]
shapes = []
for C, odim, g in units:
    S2 = 2 * (4 * g + odim)
    if C % 4 == 0:
        shapes += [(f"PQ  C={C}", Tn, S2, C, True, False), (f"dx  C={C}", Tn, C, S2, True, True), (f"dWpq C={C}", S2, C, Tn, False, True)]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, akf, bnf in shapes:
    A = torch.randn((M, K) if akf else (K, M), device="cuda")
    Bm = torch.randn((K, N) if bnf else (N, K), device="cuda")
    sam, sak = (K, 1) if akf else (1, M)
    sbk, sbn = (N, 1) if bnf else (1, K)
    Cs = {a: torch.empty((M, N), device="cuda") for a in (0, 1)}
    ts = {0: [], 1: []}
    for rnd in range(9):
        for a in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                T._gemm(A, sam, sak, Bm, sbk, sbn, Cs[a], N, None, M, N, K, a)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                ts[a].append(e0.elapsed_time(e1) / 10 * 1e3)
    same = torch.equal(Cs[0], Cs[1])
    m0, m1 = sorted(ts[0])[len(ts[0]) // 2], sorted(ts[1])[len(ts[1]) // 2]
    tot[0] += m0; tot[1] += m1
    print(f"{name:12s} [{M} x {N} x {K}]  gemm2 {m0:7.1f} us   gemm(v1) {m1:7.1f} us   {'bit-identical' if same else 'DIFFERENT'}", flush=True)
print(f"sum          gemm2 {tot[0]:7.1f} us   gemm(v1) {tot[1]:7.1f} us")
# the same shapes on the 16-bit pipe: arith 2 = split-fp16 (three products per term), 3 = split-bf16; error against float64
for name, M, N, K, akf, bnf in shapes:
    A = torch.randn((M, K) if akf else (K, M), device="cuda")
    Bm = torch.randn((K, N) if bnf else (N, K), device="cuda")
    sam, sak = (K, 1) if akf else (1, M)
    sbk, sbn = (N, 1) if bnf else (1, K)
    ref = (A.double() if akf else A.double().t()) @ (Bm.double() if bnf else Bm.double().t())
    row = f"{name:12s} [{M} x {N} x {K}] "
    for a in (0, 2, 3):
        C = torch.empty((M, N), device="cuda")
        ts = []
        for rnd in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                T._gemm(A, sam, sak, Bm, sbk, sbn, C, N, None, M, N, K, a)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        row += f"  arith {a}: {sorted(ts)[len(ts) // 2]:6.1f} us (err {err:.1e})"
    print(row, flush=True)
