#!/usr/bin/env python3
"""GPU box: randomised parity of the whole inference path against the CPU oracle - random synthetic checkpoints (seeded
synth_state_dict: non-trivial BatchNorm statistics, ActNorm, W, conditioner last layers), random patch counts / sizes / up
ratios, surface and volume clouds.  north_star's bars: kNN indices bit-exact, x within 1e-5, per-sample log-det and logp within
1e-5 relative to max(|value|, N) (they are sums of ~18 N order-1 terms that can cancel).
  python tools/stress_forward.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as O
from puflow_amd.interpflow import PointInterpFlow
from puflow_amd.weights import synth_patches, synth_state_dict

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
t_end = time.time() + budget
n = 0
worst = {"x": 0.0, "ldj": 0.0, "logp": 0.0}
while time.time() < t_end:
    wseed = int(torch.randint(0, 10 ** 6, (1,), generator=g))
    sd = synth_state_dict(wseed)
    net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.cuda().eval()
    for _ in range(3):
        B = int(torch.randint(1, 4, (1,), generator=g))
        N = [256, 320, 512, 1024, 2048][int(torch.randint(0, 5, (1,), generator=g))]
        R = [4, 4, 4, 2, 3][int(torch.randint(0, 5, (1,), generator=g))]
        x = synth_patches(B, N, seed=int(torch.randint(0, 10 ** 6, (1,), generator=g)), surface=bool(torch.randint(0, 2, (1,), generator=g)))
        ref = O.forward(sd, x, R, stages=True)
        with torch.no_grad():
            st = net.forward_stages(x.cuda(), R) if hasattr(net, "forward_stages") else None
            if st is None:
                xo, logp = net(x.cuda(), upratio=R)
                st = {"x": xo, "logp": logp}
        ex = float((st["x"].cpu() - ref["x"]).abs().max())
        # log-det and log-likelihood are sums of ~18 N terms of order 1 that may cancel (a random checkpoint can have a
        # per-sample log-det near 0): the error is taken relative to max(|value|, N), i.e. 1e-5 of a per-point term
        el = abs(float(st["logp"]) - float(ref["logp"])) / max(abs(float(ref["logp"])), float(N))
        worst["x"] = max(worst["x"], ex); worst["logp"] = max(worst["logp"], el)
        if "idx16" in st:
            assert torch.equal(st["idx16"].cpu().long(), ref["idx16"]), ("knn", wseed, B, N)
        if "ldj" in st:
            ej = float(((st["ldj"].cpu() - ref["ldj"]).abs() / ref["ldj"].abs().clamp_min(float(N))).max())
            worst["ldj"] = max(worst["ldj"], ej)
            assert ej < 1e-5, ("ldj", wseed, B, N, R, ej)
        assert ex < 1e-5 and el < 1e-5, ("parity", wseed, B, N, R, ex, el)
        n += 1
    print(f"ok {n} forwards   worst so far: max|dx| {worst['x']:.2e}  rel ldj {worst['ldj']:.2e}  rel logp {worst['logp']:.2e}", flush=True)
print(f"stress passed: {n} random (checkpoint, batch, size, ratio) forwards within 1e-5 of the oracle; worst max|dx| {worst['x']:.2e}, "
      f"rel log-det {worst['ldj']:.2e}, rel logp {worst['logp']:.2e}")
