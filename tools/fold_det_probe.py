import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict
from puflow_amd import train_ops
DEV = "cuda:0"
dense = ((synth_patches(8, 1024, seed=5) + 1) / 2).to(DEV)
batch = (dense[:, ::4].contiguous(), dense, torch.ones(8, device=DEV))
def run(dw, graphed):
    torch.manual_seed(0)
    tm = TrainerModule(default_cfg(learning_rate=1e-3, deterministic=True, dw_stream=dw), loss_mix="pugan")
    tm.network.load_state_dict(synth_state_dict(21))
    tm = tm.to(DEV).train()
    tm._sync_actnorm_init(batch)
    out = []
    opt = tm.configure_optimizers()["optimizer"]
    if graphed:
        gs = tm.graphed_train_step(batch, opt, warmup=1)
        for _ in range(3): out.append(float(gs(batch)))
    else:
        for _ in range(4): out.append(float(tm.train_step(batch, opt)))
    torch.cuda.synchronize()
    return out
print("fold", train_ops._FOLD_WU)
print("graph dw0 a", run(False, True)); print("graph dw0 b", run(False, True)); print("graph dw1  ", run(True, True))
print("eager dw0  ", run(False, False)); print("eager dw1  ", run(True, False))
