#!/usr/bin/env python3
"""GPU box: gradients of bench.py's configs[2] training step (32 x (256 -> 1024), logp + EMD + CD) with all flow blocks of a
direction as one autograd node (PF_TRAIN_CHAIN, default) against one node per block piece, from the SAME state (weights after
one optimisation step, so ActNorm is initialised), and the per-block path against itself (its run-to-run noise: float atomics
in the neighbour scatter).   python tools/chain_trajectory.py"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puflow_amd import train_ops as T
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict

dev = torch.device("cuda:0")
dense = ((synth_patches(32, 1024, seed=2021) + 1) / 2).to(dev)
batch = (dense[:, ::4].contiguous(), dense, torch.ones(32, device=dev))
tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
tm.network.load_state_dict(synth_state_dict(2021))
tm = tm.to(dev)
opt = tm.configure_optimizers()["optimizer"]
tm.train_step(batch, opt)
state = copy.deepcopy(tm.state_dict())


def grads(chain):
    T._CHAIN = chain
    tm.load_state_dict(state)
    for p in tm.parameters():
        p.grad = None
    loss = tm.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in tm.named_parameters() if p.grad is not None}


la, ga = grads(False)
la2, ga2 = grads(False)
lb, gb = grads(True)
gmax = max(float(g.abs().max()) for g in ga.values())


def worst(x, y):
    return sorted(((float((x[n] - y[n]).abs().max()) / (float(x[n].abs().max()) + 1e-5 * gmax), n) for n in x), reverse=True)[:6]


print(f"loss per-block {la:.8f} / {la2:.8f}   chain {lb:.8f}")
print("per-block vs per-block:", [(f"{v:.2e}", n) for v, n in worst(ga, ga2)])
print("per-block vs chain    :", [(f"{v:.2e}", n) for v, n in worst(ga, gb)])
for n in ("network.flow_blocks.5.coupling1.bias_net.layers.4.bias", "network.flow_blocks.0.actnorm.logs", "network.flow_blocks.3.permutate1.permutater.W"):
    print(n, ga[n].flatten()[:9].tolist(), gb[n].flatten()[:9].tolist())
