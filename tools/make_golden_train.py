#!/usr/bin/env python3
"""Golden vectors of ONE TRAINING STEP of the reference module (train() mode, BN batch statistics,
ActNorm data-dependent init) on CPU.  Loss = 1e-4*logp + 1e-1*CD(pred, dense) with a torch-CPU Chamfer
(the reference's EMD is CUDA-only and cannot run here).  Same harness shims as tools/make_golden.py.
Writes tests/golden/train_step.npz: loss pieces, x, logp, per-parameter gradient norms, a few full
gradients, BN running-stat updates and the ActNorm init values.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G      # noqa: E402  (sets PYTHONDONTWRITEBYTECODE, sys.path)
import numpy as np
import torch


def chamfer_cpu(x, y):
    d = ((x[:, :, None] - y[:, None]) ** 2).sum(-1)
    return (d.min(2)[0].mean(1) + d.min(1)[0].mean(1)).mean()


def main():
    G._install_shims()
    from puflow_amd.weights import synth_patches, synth_state_dict
    torch.set_num_threads(1)
    wseed, dseed, B, N, R = 11, 12, 4, 64, 4
    sd = synth_state_dict(wseed)
    from modules.discrete.interpflow import PointInterpFlow
    net = PointInterpFlow(3)
    net.load_state_dict(sd)
    net.train()                                       # ActNorm NOT initialised: first-batch init happens
    dense = synth_patches(B, N * R, seed=dseed)
    sparse = dense[:, ::R].contiguous()
    x, logp = net(sparse, R)
    cd = chamfer_cpu(x, dense)
    loss = logp * 1e-4 + cd * 1e-1
    loss.backward()
    out = {"meta_wseed": np.int64(wseed), "meta_dseed": np.int64(dseed), "meta_B": np.int64(B), "meta_N": np.int64(N),
           "x": x.detach().numpy(), "logp": logp.detach().numpy(), "cd": cd.detach().numpy(), "loss": loss.detach().numpy()}
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(0.0 if p.grad is None else float(p.grad.norm()))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, np.float64)
    for k in ("feat_convs.0.convs.0.0.weight", "feat_convs.3.conv_out.weight", "merge_convs.2.conv2.weight",
              "flow_blocks.1.permutate1.permutater.W", "flow_blocks.0.actnorm.logs", "flow_blocks.4.coupling2.scale_net.layers.4.weight",
              "interp.weight_unit.mlp.6.weight", "interp.knn_context.feat_conv.convs.3.1.weight"):
        out["grad::" + k] = dict(net.named_parameters())[k].grad.numpy()
    sd2 = net.state_dict()
    for k in ("feat_convs.2.convs.1.1.running_mean", "feat_convs.2.convs.1.1.running_var", "interp.weight_unit.mlp.1.running_var",
              "flow_blocks.0.actnorm.logs", "flow_blocks.0.actnorm.bias", "flow_blocks.5.actnorm.logs"):
        out["state::" + k] = sd2[k].numpy()
    path = os.path.join(G.ROOT, "tests", "golden", "train_step.npz")
    np.savez_compressed(path, **out)
    print("loss", float(loss), "logp", float(logp), "cd", float(cd), "size", os.path.getsize(path))


if __name__ == "__main__":
    main()
