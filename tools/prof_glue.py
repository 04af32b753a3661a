#!/usr/bin/env python3
"""GPU box: which Python lines of one eager training step launch torch's own kernels (copies, adds, cats, fills) - the glue left
between the HIP kernels.  python tools/prof_glue.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict

DEV = "cuda:0"
dense = ((synth_patches(32, 1024, seed=5) + 1) / 2).to(DEV)
batch = (dense[:, ::4].contiguous(), dense, torch.ones(32, device=DEV))
torch.manual_seed(0)
tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
tm.network.load_state_dict(synth_state_dict(21))
tm = tm.to(DEV)
opt = tm.configure_optimizers()["optimizer"]
for _ in range(3):
    tm.train_step(batch, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tm.train_step(batch, opt)
    torch.cuda.synchronize()
rows = collections.Counter()
shapes = {}
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::mul", "aten::div", "aten::sum", "aten::index", "aten::where", "aten::stack"):
        continue
    if not ev.kernels and ev.name not in ("aten::cat", "aten::add", "aten::copy_"):
        pass
    st = [s for s in ev.stack if "/puflow_amd/" in s or "/repo/" in s]
    key = (ev.name, st[0].strip() if st else "(autograd engine)", str(ev.input_shapes)[:70])
    rows[key] += 1
for (name, where, shp), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:16s} {where[-95:]:95s} {shp}")
