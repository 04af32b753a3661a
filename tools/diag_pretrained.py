#!/usr/bin/env python3
"""GPU box: per-stage error of the HIP path against the pretrained-weights golden, per EdgeConv arithmetic mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches
g = np.load("tests/golden/pretrained_pu1k.npz")
sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.cuda().eval()
for case in ("a", "b"):
    B, N, seed = (int(v) for v in g[f"{case}/meta"])
    xyz = synth_patches(B, N, seed=seed, surface=True).cuda()
    for mode in ("f16n", "f32"):
        net.ec_mode = mode
        st = net.forward_stages(xyz, 4)
        errs = {}
        for k in ("cs0", "cs5"):
            i = int(k[2:]); ref = g[f"{case}/{k}"]; n = ref.shape[-1]
            errs[k] = float(np.abs(st["cs"][i].cpu().numpy()[..., :n] - ref).max()), float(np.abs(ref).max())
        for k in ("z", "fz", "x"):
            ref = g[f"{case}/{k}"]
            errs[k] = float(np.abs(st[k].cpu().numpy() - ref).max()), float(np.abs(ref).max())
        errs["ldj_rel"] = float(np.abs((st["ldj"].cpu().numpy() - g[f"{case}/ldj"]) / g[f"{case}/ldj"]).max())
        print(case, mode, {k: (f"{v[0]:.2e}/{v[1]:.1f}" if isinstance(v, tuple) else f"{v:.2e}") for k, v in errs.items()}, flush=True)
