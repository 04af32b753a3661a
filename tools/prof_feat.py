#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: forward + backward of the six feature-extractor units (training mode,
32 x 256 points), 5 repetitions after warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import train_ops, ops
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(2021)); net = net.cuda().train()
xyz = ((synth_patches(B, 1024, seed=2021) + 1) / 2).cuda()[:, ::4].contiguous()
idx16, _ = ops.knn_idx32(xyz, xyz, 16)
for it in range(7):
    for p in net.parameters():
        p.grad = None
    h, outs = xyz, []
    for i in range(net.num_blocks):
        h = train_ops.edgeconv_train(net.feat_convs[i], h, idx16)
        outs.append(h)
    sum(o.sum() for o in outs).backward()
torch.cuda.synchronize()
