#!/usr/bin/env python3
"""GPU box: a few EAGER training steps of BASELINE configs[2] (32 x (256 -> 1024), logp + EMD + CD, clip, Adam) - the target
of the PMC passes for the training kernels (counters are attributed per kernel launch; a graph replay is one dispatch group):
  bash tools/pmc_cmd.sh <tag> tools/train_eager_steps.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
dense = ((synth_patches(32, 1024, seed=2021) + 1) / 2).to(dev)
batch = (dense[:, ::4].contiguous(), dense, torch.ones(32, device=dev))
tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
tm.network.load_state_dict(synth_state_dict(2021))
tm = tm.to(dev)
opt = tm.configure_optimizers()["optimizer"]
for _ in range(steps):
    loss = tm.train_step(batch, opt)
torch.cuda.synchronize()
print("loss", float(loss))
