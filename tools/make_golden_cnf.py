#!/usr/bin/env python3
"""Golden vectors for the continuous model's ODE right-hand side, produced by the REFERENCE's own code.

Runs ONLY in the build container (needs /root/reference).  `modules/continuous/odefunc.py` and `diffeq_layers.py`
import without torchdiffeq (only `cnf.py` needs it), so the network + Hutchinson divergence of a CNF block - everything
except the ODE solver - can be pinned: `ODEfunc.forward(t, (y, logp, c))` with the noise fixed through
`before_odeint_forward(e)` / `before_odeint_inverse(upratio)` (odefunc.py:114-148).
Writes tests/golden/cnf_rhs.npz (inputs + outputs; weights come from puflow_amd.weights.synth_cnf_state_dict(seed)).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from modules.continuous.odefunc import ODEfunc, ODEnet          # noqa: E402  (the reference)
from puflow_amd.weights import CNF_COND_CHANNELS, synth_cnf_state_dict   # noqa: E402

SEED = 11


def main():
    sd = synth_cnf_state_dict(SEED)
    out = {"meta_seed": np.int64(SEED)}
    g = torch.Generator().manual_seed(SEED)
    for block, (B, N), R in ((0, (2, 40), 1), (3, (1, 96), 1), (5, (2, 24), 4), (2, (1, 50), 4)):
        cd = CNF_COND_CHANNELS[block]
        f = ODEfunc(ODEnet((64, 64), input_shape=(3,), context_dim=cd, layer_type="concatsquash", nonlinearity="tanh"))
        pfx = f"flow_blocks.{block}.cnf.odefunc."
        missing = f.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}, strict=True)
        e = torch.randn(B, N, 3, generator=g)
        c = torch.randn(B, N, cd, generator=g) * 0.7
        t = torch.tensor(0.05 + 0.11 * block)
        f.before_odeint_forward(e.clone())
        if R > 1:                                                       # the inverse pass: states of N*R rows, e repeated
            f.before_odeint_inverse(R)
            c_in = torch.repeat_interleave(c, R, dim=1)
        else:
            c_in = c
        y = torch.randn(B, N * R, 3, generator=g) * 0.8
        logp = torch.zeros(B, N * R, 1)
        dy, ndiv, dc = f(t, (y, logp, c_in))
        assert float(dc.detach().abs().max()) == 0.0
        tag = f"b{block}_R{R}"
        out[f"{tag}_t"] = np.float32(t.item())
        for name, val in (("y", y), ("c", c), ("e", e), ("dy", dy.detach()), ("ndiv", ndiv.detach())):
            out[f"{tag}_{name}"] = val.detach().numpy().astype(np.float32)
        print(tag, "dy", float(dy.abs().max()), "ndiv", float(ndiv.abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "cnf_rhs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
