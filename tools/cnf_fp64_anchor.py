#!/usr/bin/env python3
"""CPU only (build container or the GPU box's host cores): how well-determined is a synthetic CNF workload?  For each profile
the oracle (oracle/cnf_ref.py) runs an item alone and inside a batch of two, in float32 and float64:
  * `f32 - f64`        : the fp32 oracle's own rounding error at an identical step sequence (the yardstick for a kernel);
  * `step sequence`    : float64 item-alone vs float64 item-in-a-batch - the RMS error norm runs over the batch, so the steps differ
                         and the two float64 solutions differ by what the solver's rtol = 1e-5 leaves undetermined.
This is the evidence behind DESIGN ledger 1a (why `dynamics = 5, T = 0.5` was replaced by the trained checkpoint's end times).
  python tools/cnf_fp64_anchor.py [N]  > profiles/r4_final/cnf_fp64_anchor.txt"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cnf_ref as C
from puflow_amd.weights import CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES, synth_cnf_state_dict, synth_patches

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.set_num_threads(min(8, os.cpu_count() or 1))


def trial(name, sd):
    xyz = synth_patches(2, N, seed=2021)
    g = torch.Generator().manual_seed(0)
    noise = [torch.randn(2, N, 3, generator=g) for _ in range(6)]
    t = time.time()
    full = C.forward(sd, xyz, 4, noise=noise, stages=True, dtype=torch.float64)
    one = C.forward(sd, xyz[:1], 4, noise=[n[:1] for n in noise], stages=True, dtype=torch.float64)
    o32 = C.forward(sd, xyz[:1], 4, noise=[n[:1] for n in noise], stages=True)
    d = (full["x"][:1] - one["x"]).abs().max(-1)[0].flatten()
    print(f"{name:34s} nfe {one['nfe']:4d} acc {one['accepted']:3d} rej {one['rejected']:2d} (f32: {o32['nfe']}/{o32['rejected']}) | "
          f"f32 - f64: x {float((o32['x'].double() - one['x']).abs().max()):.2e} z {float((o32['z'].double() - one['z']).abs().max()):.2e} | "
          f"step sequence (f64): x max {float(d.max()):.2e} median {float(d.median()):.2e} z {float((full['z'][:1] - one['z']).abs().max()):.2e} | "
          f"max|x| {float(one['x'].abs().max()):.2f} max|z| {float(one['z'].abs().max()):.2f}  [{time.time() - t:.1f} s]", flush=True)


print(f"oracle/cnf_ref.py, 1 x {N} points (and the same item inside a batch of 2), x4, seeds of bench.py --mode cnf")
trial("round 3: dynamics 5, T = 0.5", synth_cnf_state_dict(2021, dynamics=5.0))
trial("random init: dynamics 1, T = 0.5", synth_cnf_state_dict(2021, dynamics=1.0))
trial("dynamics 1, trained end times", synth_cnf_state_dict(2021, dynamics=1.0, end_times=CNF_PU1K_END_TIMES))
trial("round 4: dynamics 1.7, trained T", synth_cnf_state_dict(2021, dynamics=CNF_PU1K_DYNAMICS, end_times=CNF_PU1K_END_TIMES))
trial("dynamics 2, trained end times", synth_cnf_state_dict(2021, dynamics=2.0, end_times=CNF_PU1K_END_TIMES))
