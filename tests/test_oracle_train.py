"""Train-mode oracle (BN batch statistics, ActNorm init, gradients by torch autograd on the functional
restatement) against a training step of the reference itself (tools/make_golden_train.py)."""
import os

import numpy as np
import torch

from oracle import ref_cpu as O
from puflow_amd.weights import synth_patches, synth_state_dict


def _chamfer(x, y):
    d = ((x[:, :, None] - y[:, None]) ** 2).sum(-1)
    return (d.min(2)[0].mean(1) + d.min(1)[0].mean(1)).mean()


def test_train_step_matches_reference(golden_dir):
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    B, N, R = int(g["meta_B"]), int(g["meta_N"]), 4
    sd = synth_state_dict(int(g["meta_wseed"]))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    dense = synth_patches(B, N * R, seed=int(g["meta_dseed"]))
    sparse = dense[:, ::R].contiguous()
    x, logp, ts = O.forward_train(sd, sparse, R, actnorm_init=True)
    cd = _chamfer(x, dense)
    loss = logp * 1e-4 + cd * 1e-1
    loss.backward()
    np.testing.assert_allclose(x.detach().numpy(), g["x"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(float(logp), float(g["logp"]), rtol=1e-6)
    np.testing.assert_allclose(float(cd), float(g["cd"]), rtol=1e-5)
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    floor = 1e-6 * max(norms.values())       # conv biases in front of a batch-stat BN have an exactly-zero gradient:
    for k, ref in norms.items():             # both sides then hold rounding noise only
        got = 0.0 if sd[k].grad is None else float(sd[k].grad.norm())
        assert abs(got - ref) <= 1e-4 * ref + floor, (k, got, ref)
    for key in g.files:
        if key.startswith("grad::"):
            np.testing.assert_allclose(sd[key[6:]].grad.numpy(), g[key], rtol=2e-4, atol=floor)
        if key.startswith("state::"):
            k = key[7:]
            v = ts.bn_updates.get(k, ts.actnorm_init.get(k))
            np.testing.assert_allclose(v.detach().numpy(), g[key], rtol=1e-5, atol=1e-7)
