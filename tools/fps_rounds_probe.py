#!/usr/bin/env python3
"""GPU box: exchange rounds and time of the FPS merge for the CLI's clouds (the merge input of PatchHelper.upsample), per cloud,
with and without the layout hint."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops, _lib
from puflow_amd.patch import PatchHelper
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict
dev = "cuda:0"
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.to(dev).eval()
ph = PatchHelper(256, 4)
lib = _lib.load()
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for k in range(NC):
    pc = synth_patches(1, 5000, seed=100 + k).to(dev)
    with torch.no_grad():
        pcn, gc, gfd = PatchHelper.normalize_pc(pc)
        patches = PatchHelper.extract_knn_patch(pcn, ph.knn, 256, 4)
        cand = PatchHelper.upsampling_patches(net, patches, 4)
    M = cand.shape[1] * cand.shape[2]
    flat = cand.reshape(1, M, 3).contiguous()
    out = []
    for grp in (0, cand.shape[2]):
        mind = torch.empty((1, M), dtype=torch.float32, device=dev)
        idx = torch.zeros((1, 20024), dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.pf_fps_grouped(flat.data_ptr(), 1, M, 20024, grp, mind.data_ptr(), idx.data_ptr(), s), "fps")
            b.record(); torch.cuda.synchronize()
        stride, word = ctypes.c_longlong(0), ctypes.c_longlong(0)
        lib.pf_fps_scratch_layout(M, ctypes.byref(stride), ctypes.byref(word))
        rounds = int(mind.view(-1)[: M // 2 * 2].view(torch.int64)[word.value + 1].item())
        out.append((grp, rounds, round(a.elapsed_time(b), 2)))
    print(f"cloud {k}: " + "   ".join(f"group {g}: {r} rounds {t} ms" for g, r, t in out), flush=True)
