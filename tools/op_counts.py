#!/usr/bin/env python3
"""GPU box: which torch operators still launch kernels in one eager training step (count and device time per operator)."""
import os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tm.train_step(batch, opt)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[1])
for k, c, t in rows[:45]:
    print(f"{k[:70]:70s} {c:5d} {t:9.1f} us")
