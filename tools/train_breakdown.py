#!/usr/bin/env python3
"""GPU box: where the training step's time goes (BASELINE configs[2] shape, 32 x 256 -> 1024).  Sections are timed
eagerly with a device sync on both sides: forward and backward of each part of the network separately (the part's
outputs summed as a stand-in loss), the losses, and clip + Adam.      python tools/train_breakdown.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import train_ops, ops
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
for _ in range(3):
    tm.train_step(batch, opt)
net = tm.network


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def count_launches(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    return sum(e.count for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA)


idx16, _ = ops.knn_idx32(sparse, sparse, 16)
res = {}


def feat_fwd():
    h = sparse
    outs = []
    for i in range(net.num_blocks):
        h = train_ops.edgeconv_train(net.feat_convs[i], h, idx16)
        outs.append(h)
    return outs


def feat_fb():
    for p in net.parameters():
        p.grad = None
    sum(o.sum() for o in feat_fwd()).backward()


def full_fwd():
    return train_ops.forward_train(net, sparse, 4)


def full_fb():
    for p in net.parameters():
        p.grad = None
    u, logp = full_fwd()
    (u.sum() + logp).backward()


def loss_fb():
    for p in net.parameters():
        p.grad = None
    tm.training_step(batch, 0).backward()


def opt_only():
    torch.nn.utils.clip_grad_norm_(tm.parameters(), 1e-2, foreach=True)
    opt.step()


with torch.no_grad():
    res["feature extractor fwd (no grad graph)"] = timed(feat_fwd)
res["feature extractor fwd"] = timed(feat_fwd)
res["feature extractor fwd+bwd"] = timed(feat_fb)
res["network fwd"] = timed(full_fwd)
res["network fwd+bwd"] = timed(full_fb)
res["network + losses fwd+bwd"] = timed(loss_fb)
res["clip + Adam"] = timed(opt_only)
res["train_step"] = timed(lambda: tm.train_step(batch, opt))
for k, v in res.items():
    print(f"{k:42s} {v:8.2f} ms", flush=True)
try:
    print("launches: feature fwd+bwd", count_launches(feat_fb), " network fwd+bwd", count_launches(full_fb),
          " step", count_launches(lambda: tm.train_step(batch, opt)), flush=True)
except Exception as ex:                                     # the profiler is optional
    print("launch count unavailable:", ex)
