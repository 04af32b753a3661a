#!/usr/bin/env python3
"""GPU box: does the FPS merge kernel slow down when the network's launches (graph replay + side stream) precede it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops
from puflow_amd.patch import PatchHelper
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict
dev = "cuda:0"
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net.set_to_initialized_state(); net = net.to(dev).eval()
ph = PatchHelper(256, 4)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
    mode_threads = int(sys.argv[2])
if mode == "shuffle_nogc":
    import gc; gc.disable()
if mode == "shuffle_gcstat":
    import gc
    gc.callbacks.append(lambda phase, info: print("   gc", phase, info, flush=True) if phase == "stop" and info.get("generation", 0) >= 1 else None)
ts = []
HOST = []
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(max_workers=1)
import numpy as np, tempfile
tmpd = tempfile.mkdtemp()
from puflow_amd.upsample import save_xyz
def burn_py(ms):
    t0 = time.perf_counter(); x = 0
    while (time.perf_counter() - t0) * 1e3 < ms: x += 1
    return x
for k in range(24):
    pc = synth_patches(1, 5000, seed=100 + k)
    if mode.startswith("shuffle"):
        torch.manual_seed(2021 + k); pc = pc[:, torch.randperm(5000)].contiguous()
    pc = pc.to(dev)
    with torch.no_grad():
        pcn, gc, gfd = PatchHelper.normalize_pc(pc)
        patches = PatchHelper.extract_knn_patch(pcn, ph.knn, 256, 4)
        cand = PatchHelper.upsampling_patches(net, patches, 4)
        if mode == "sync":
            torch.cuda.synchronize()
        flat = cand.reshape(1, -1, 3).contiguous()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if mode == "shuffle_fixed":
            if k == 0:
                from puflow_amd import _lib
                FM = torch.empty((1, flat.shape[1]), dtype=torch.float32, device=dev); FI = torch.zeros((1, 20024), dtype=torch.int32, device=dev)
            _lib.check(_lib.load().pf_fps_grouped(flat.data_ptr(), 1, flat.shape[1], 20024, cand.shape[2], FM.data_ptr(), FI.data_ptr(), torch.cuda.current_stream().cuda_stream), "fps")
            idx = FI
        elif mode in ("shuffle_v1", "shuffle_v2", "shuffle_v3", "shuffle_v4", "shuffle_v5a", "shuffle_v5b", "shuffle_v5c"):
            from puflow_amd import _lib
            def call():
                mind = torch.empty((1, flat.shape[1]), dtype=torch.float32, device=dev)
                idx = torch.zeros((1, 20024), dtype=torch.int32, device=dev)
                _lib.check(_lib.load().pf_fps_grouped(flat.data_ptr(), 1, flat.shape[1], 20024, cand.shape[2], mind.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream), "fps")
                if mode.startswith("shuffle_v5"):
                    torch.cuda.synchronize()
                    words = mind.view(-1)[: flat.shape[1] // 2 * 2].view(torch.int64)
                    if mode == "shuffle_v5a":
                        pos = torch.arange(1, device=dev, dtype=torch.int64) * 49920 + 2048
                        x = words[pos]
                    if mode == "shuffle_v5b":
                        x = bool((words[2048:2049] != 0).any())
                    if mode == "shuffle_v5c":
                        x = int(words[2048].item())
                if mode in ("shuffle_v2", "shuffle_v3", "shuffle_v4"):
                    t0 = time.perf_counter()
                    if mode == "shuffle_v3": torch.cuda.current_stream().synchronize()
                    if mode == "shuffle_v4": torch.cuda.synchronize()
                    ops._check_fps_abort(_lib.load(), mind, 1, flat.shape[1])
                    HOST.append(round((time.perf_counter() - t0) * 1e3, 1))
                    print("   mind", hex(mind.data_ptr()), "idx", hex(idx.data_ptr()), "flat", hex(flat.data_ptr()), HOST[-1], flush=True)
                return idx
            idx = call()
        elif mode == "shuffle_mind":
            from puflow_amd import _lib
            mind = torch.empty((1, flat.shape[1]), dtype=torch.float32, device=dev)
            idx = torch.zeros((1, 20024), dtype=torch.int32, device=dev)
            _lib.check(_lib.load().pf_fps_grouped(flat.data_ptr(), 1, flat.shape[1], 20024, cand.shape[2], mind.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream), "fps")
            MIND = mind.data_ptr()
        else:
            idx = ops.furthest_point_sample(flat, 20024, group=cand.shape[2])
        b.record(); torch.cuda.synchronize()
        ts.append(round(a.elapsed_time(b), 1))
        if mode == "shuffle_mind":
            print(k, ts[-1], "mind", hex(MIND), "flat", hex(flat.data_ptr()), "idx", hex(idx.data_ptr()), "cand", hex(cand.data_ptr()), flush=True)
        if mode == "shuffle_addr":
            print(k, ts[-1], hex(flat.data_ptr()), hex(idx.data_ptr()), flush=True)
        if mode == "cpu":
            out = flat[0, idx[0].long()].cpu()
        if mode == "thread_py":
            pool.submit(burn_py, 30.0)
        if mode == "thread_save":
            out = flat[0, idx[0].long()].cpu().numpy()
            pool.submit(save_xyz, os.path.join(tmpd, f"c{k}.xyz"), out)
        if mode == "outl":
            den = flat[:, idx[0].long()].contiguous()
            pred = PatchHelper.remove_outliers(den, pc, 24).cpu().numpy()
print(mode, ts)
if HOST: print('host ms in _check_fps_abort', HOST)
