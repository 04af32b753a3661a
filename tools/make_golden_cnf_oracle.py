#!/usr/bin/env python3
"""CPU, build container: cache what the continuous model's CPU oracle (oracle/cnf_ref.py, fp32 and float64) returns on the
inputs of the two slowest GPU tests, so that the GPU suite does not spend a minute of every round re-running a CPU oracle
(tests/test_gpu_cnf.py::test_full_size_properties_32x2048 alone was 51 s, mostly this).

    python tools/make_golden_cnf_oracle.py        -> tests/golden/cnf_oracle_cache.npz

The fixture holds ORACLE outputs (test infrastructure talking to itself), not reference outputs: the oracle's own pinning to the
reference is tests/test_oracle_cnf.py.  The tests check a fingerprint of their inputs against the one stored here and fall back
to running the oracle when it does not match (changed seeds, changed synthetic weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import cnf_ref as C
from puflow_amd.weights import CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES, synth_cnf_state_dict, synth_patches


def fingerprint(xyz, noise):
    return np.array([float(xyz.double().sum()), float(xyz.double().abs().sum())] + [float(n.double().sum()) for n in noise])


def pack(prefix, o, keys, out, f64=False):
    for k in keys:
        v = o[k]
        out[f"{prefix}_{k}"] = (v.double() if f64 else v).numpy() if isinstance(v, torch.Tensor) else np.array(v)


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    out = {}
    # ---- test_full_size_properties_32x2048: items 0..1 of the bench workload
    sd = synth_cnf_state_dict(2021, dynamics=CNF_PU1K_DYNAMICS, end_times=CNF_PU1K_END_TIMES)
    xyz = synth_patches(32, 2048, seed=2021)
    g = torch.Generator().manual_seed(0)
    noise = [torch.randn(32, 2048, 3, generator=g) for _ in range(6)]
    n2, n1 = [n[:2] for n in noise], [n[:1] for n in noise]
    out["full_fp"] = fingerprint(xyz[:2], n2)
    o32 = C.forward(sd, xyz[:2], 4, noise=n2, stages=True)
    o64 = C.forward(sd, xyz[:2], 4, noise=n2, stages=True, dtype=torch.float64)
    o64_1 = C.forward(sd, xyz[:1], 4, noise=n1, stages=True, dtype=torch.float64)
    pack("full_o32", o32, ("x", "z", "ldj", "nfe", "accepted", "rejected"), out)
    pack("full_o64", o64, ("x", "z", "ldj", "nfe", "accepted", "rejected"), out, f64=True)
    pack("full_o64_1", o64_1, ("x",), out, f64=True)
    # ---- test_pretrained_checkpoint_forward_against_the_fp64_anchor
    gp = np.load(os.path.join(ROOT, "tests", "golden", "pretrained_cnf.npz"))
    sdp = {k[3:]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("sd/")}
    xyzp = torch.from_numpy(gp["xyz"])
    noisep = [torch.from_numpy(n) for n in gp["noise"]]
    out["pre_fp"] = fingerprint(xyzp, noisep)
    p32 = C.forward(sdp, xyzp, 4, noise=noisep, stages=True)
    p64 = C.forward(sdp, xyzp, 4, noise=noisep, stages=True, dtype=torch.float64)
    pack("pre_o32", p32, ("x", "z", "ldj", "idx16", "nfe", "accepted", "rejected"), out)
    pack("pre_o64", p64, ("x", "z", "ldj", "nfe", "accepted", "rejected"), out, f64=True)
    path = os.path.join(ROOT, "tests", "golden", "cnf_oracle_cache.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
