#!/usr/bin/env python3
"""GPU box: where the remaining small torch launches of a training step come from (operator + Python call site)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict
dev = "cuda:0"
dense = ((synth_patches(32, 1024, seed=2021) + 1) / 2).to(dev)
sparse = dense[:, ::4].contiguous()
batch = (sparse, dense, torch.ones(32, device=dev))
tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
tm.network.load_state_dict(synth_state_dict(2021))
tm = tm.to(dev)
opt = tm.configure_optimizers()["optimizer"]
for _ in range(3):
    tm.train_step(batch, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tm.train_step(batch, opt)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros",
        "aten::zeros_like", "aten::full", "aten::mul", "aten::div", "aten::neg", "aten::cat", "aten::sum")
cnt = collections.Counter()
for e in prof.events():
    if e.name in want and e.device_time_total > 0 or (e.name in ("aten::fill_", "aten::add_", "aten::copy_") and e.device_time_total > 0):
        site = "<autograd / C++>"
        for fr in e.stack:
            if "/root/repo" in fr or "puflow_amd" in fr or "tools/" in fr:
                site = fr.split("/")[-1][:70]
                break
        if site.startswith("<"):
            site += " " + str([tuple(x) for x in (e.input_shapes or []) if x])[:60]      # autograd's own adds: the shapes name the tensor
        cnt[(e.name, site)] += 1
for (name, site), c in cnt.most_common(60):
    print(f"{c:5d}  {name:18s} {site}")
