#!/usr/bin/env python3
"""GPU box: the batched weight-gradient launch of the flow chain's conditioner MLPs (pf_mlp_train_dw_batch: mlp_dw_kernel +
mlp_dw_reduce_kernel) alone, on the three shapes of the training step (32 x 256 points): g direction (6 nets x 32768 rows),
f direction (6 x 8192), the 12 scale / shift nets (12 x 8192).  HIP events; `chunk` sweep.   python tools/time_mlpdw.py [chunks...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
f32 = dict(dtype=torch.float32, device=dev)
T = 8192
CCS = [32, 64, 128, 128, 128, 128]


def build(kind, chunk):
    descs, keep = [], []
    if kind in ("g", "f"):
        R = 4 if kind == "g" else 1
        rows = T * R
        for i, cc in enumerate(CCS):
            td = 1 if i % 2 == 0 else 2
            d = _lib.PfMlpTrain()
            d.rows, d.nl, d.td, d.cc, d.cdiv, d.ldy = rows, 3, td, cc, R, 3
            w = [64, 64, 3 - td]
            ins = [cc + td, 64, 64]
            y = torch.randn(rows, 3, **f32); c = torch.randn(T, cc, **f32)
            h = [torch.randn(rows, 64, **f32) for _ in range(2)]
            dz = [torch.randn(rows, 64, **f32) for _ in range(2)]
            dout = torch.randn(rows, w[2], **f32)
            dW = [torch.empty(w[l], ins[l], **f32) for l in range(3)]
            db = [torch.empty(w[l], **f32) for l in range(3)]
            keep += [y, c, *h, *dz, dout, *dW, *db]
            for l in range(3):
                d.width[l] = w[l]; d.dW[l] = dW[l].data_ptr(); d.db[l] = db[l].data_ptr(); d.W[l] = dW[l].data_ptr()
            d.y, d.c = y.data_ptr(), c.data_ptr()
            d.h[0], d.h[1], d.dz[0], d.dz[1], d.dout = h[0].data_ptr(), h[1].data_ptr(), dz[0].data_ptr(), dz[1].data_ptr(), dout.data_ptr()
            d.chunk = chunk
            if kind == "g" and os.environ.get("PF_TIME_DZSUM", "1") != "0":      # as pf_flowchain_bwd hands it over: dc = replica-summed dz[0]
                dzs = dz[0].view(T, R, 64).sum(1).contiguous()
                keep.append(dzs)
                d.dc, d.flags = dzs.data_ptr(), 1
            descs.append(d)
    else:
        for k in range(12):
            cc = CCS[k // 2]
            d = _lib.PfMlpTrain()
            d.rows, d.nl, d.td, d.cc, d.cdiv, d.ldy = T, 3, 0, cc, 1, 0
            w = [64, 64, 3]; ins = [cc, 64, 64]
            c = torch.randn(T, cc, **f32)
            h = [torch.randn(T, 64, **f32) for _ in range(2)]
            dz = [torch.randn(T, 64, **f32) for _ in range(2)]
            dout = torch.randn(T, 3, **f32)
            dW = [torch.empty(w[l], ins[l], **f32) for l in range(3)]
            db = [torch.empty(w[l], **f32) for l in range(3)]
            keep += [c, *h, *dz, dout, *dW, *db]
            for l in range(3):
                d.width[l] = w[l]; d.dW[l] = dW[l].data_ptr(); d.db[l] = db[l].data_ptr(); d.W[l] = dW[l].data_ptr()
            d.c = c.data_ptr()
            d.h[0], d.h[1], d.dz[0], d.dz[1], d.dout = h[0].data_ptr(), h[1].data_ptr(), dz[0].data_ptr(), dz[1].data_ptr(), dout.data_ptr()
            d.chunk = chunk
            descs.append(d)
    n = len(descs)
    arr = (_lib.PfMlpTrain * n)(*descs)
    need = [lib.pf_mlp_train_ws_floats(ctypes.byref(arr[k])) for k in range(n)]
    ws = torch.empty(sum(need), **f32)
    off = 0
    for k in range(n):
        arr[k].ws, arr[k].ws_floats = ws.data_ptr() + 4 * off, need[k]
        off += need[k]
    keep.append(ws)
    return arr, n, keep, sum(need)


dd = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
chunks = [int(a) for a in sys.argv[1:]] or [0]
for kind in ("g", "f", "cond"):
    for chunk in chunks:
        ch = chunk or {"g": 512, "f": 256, "cond": 256}[kind]           # what the training step uses
        arr, n, keep, wsf = build(kind, ch)
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(lib.pf_mlp_train_dw_batch(arr, n, dd.data_ptr(), s), "dw")
            e1.record(); torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        print(f"{kind:5s} chunk {ch:5d}: {sorted(ts)[len(ts) // 2]:7.1f} us   (partials {wsf * 4 / 1e6:.1f} MB)", flush=True)
        del arr, keep
