#!/usr/bin/env python3
"""GPU box: BASELINE configs[2] training step (32 x (256 -> 1024) patches, loss 1e-4 logp + 5e-2 EMD + 1e-1 CD, clip, Adam)
eager vs captured in a hipGraph; checks that both take the same optimisation trajectory.  python tools/time_train.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.trainer import TrainerModule, default_cfg
from puflow_amd.weights import synth_patches, synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
dense = ((synth_patches(B, 1024, seed=2021) + 1) / 2).to(dev)
sparse = dense[:, ::4].contiguous()
batch = (sparse, dense, torch.ones(B, device=dev))


def make():
    torch.manual_seed(0)
    tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
    tm.network.load_state_dict(synth_state_dict(2021))
    tm = tm.to(dev)
    return tm, tm.configure_optimizers()["optimizer"]


def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


tm, opt = make()
for _ in range(3):
    tm.train_step(batch, opt)
ms_e, loss_e = timeit(lambda: tm.train_step(batch, opt), 10)
print(f"eager   : {ms_e:8.2f} ms / step   loss {float(loss_e):.6f}", flush=True)

tm2, opt2 = make()
step = tm2.graphed_train_step(batch, opt2)
ms_g, loss_g = timeit(lambda: step(batch), 20)
print(f"graphed : {ms_g:8.2f} ms / step   loss {float(loss_g):.6f}   = {B / ms_g * 1e3:.0f} patches/s", flush=True)

# same trajectory: N eager steps vs N graphed steps from the same start
tm3, opt3 = make()
tm4, opt4 = make()
g4 = tm4.graphed_train_step(batch, opt4, ) if False else None
for _ in range(6):
    le = tm3.train_step(batch, opt3)
tm5, opt5 = make()
st5 = tm5.graphed_train_step(batch, opt5, )
# the graphed object ran `warmup` eager-equivalent steps during construction (2) - run 4 more
for _ in range(4):
    lg = st5(batch)
w_e = tm3.network.feat_convs[3].conv_out.weight.detach()
w_g = tm5.network.feat_convs[3].conv_out.weight.detach()
print(f"after 6 steps: eager loss {float(le):.6f}  graphed loss {float(lg):.6f}  max|dW| {float((w_e - w_g).abs().max()):.3e}", flush=True)
