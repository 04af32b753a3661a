#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: a few eager training steps (BASELINE configs[2] shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
dense = ((synth_patches(B, 1024, seed=2021) + 1) / 2).to(dev)
sparse = dense[:, ::4].contiguous()
batch = (sparse, dense, torch.ones(B, device=dev))
tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
tm.network.load_state_dict(synth_state_dict(2021))
tm = tm.to(dev)
opt = tm.configure_optimizers()["optimizer"]
for _ in range(6):
    tm.train_step(batch, opt)
torch.cuda.synchronize()
