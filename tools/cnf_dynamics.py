#!/usr/bin/env python3
"""GPU box: solver work of the continuous model as a function of the synthetic ODE nets' scale (weights.synth_cnf_state_dict
`dynamics`): function evaluations, accepted / rejected steps, ms per forward.  bench.py --mode cnf uses the scale at which
dopri5 works as hard as on the reference's pretrained checkpoint (462 evaluations, 62 accepted / 11 rejected: DESIGN 9).
python tools/cnf_dynamics.py [B] [N] [scale ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd.cnf import PointInterpFlow
from puflow_amd.weights import synth_cnf_state_dict, synth_patches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
scales = [float(v) for v in sys.argv[3:]] or [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
xyz = synth_patches(B, N, seed=2021).cuda()
torch.manual_seed(0)
noise = [torch.randn(B, N, 3, device="cuda") for _ in range(6)]
for sc in scales:
    net = PointInterpFlow(3); net.load_state_dict(synth_cnf_state_dict(2021, dynamics=sc)); net = net.cuda().eval()
    try:
        net(xyz, 4, noise=noise)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, logp = net(xyz, 4, noise=noise)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"dynamics {sc:4.1f}: {net.last_stats}  {el * 1e3:8.2f} ms  finite {bool(torch.isfinite(x).all())}  |x|max {float(x.abs().max()):.3f}", flush=True)
    except Exception as ex:
        print(f"dynamics {sc:4.1f}: {type(ex).__name__}: {str(ex)[:120]}", flush=True)
