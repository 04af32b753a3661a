#!/usr/bin/env python3
"""tests/golden/pretrained_cnf.npz: the reference's TRAINED continuous checkpoint (`pretrain/puflow-x4-cnf-pu1k.pt`) as
test data.  Build container only (needs /root/reference, read-only).

The fixture is data:
  * `sd/<key>`: the checkpoint's 390 tensors (an input of the computation - the GPU box has no /root/reference);
  * `xyz`, `noise`: one seeded 256-point surface patch and the six Hutchinson vectors of a forward (seeded; the reference
    draws them with torch.randn_like, odefunc.py:134-137);
  * `rhs/b<i>_R<r>_{t,y,c,e,dy,ndiv}`: for every flow block, what the REFERENCE's own `ODEfunc.forward` returns with the
    trained weights (network + Hutchinson divergence; `modules/continuous/odefunc.py` and `diffeq_layers.py` import without
    torchdiffeq) on states along the trained model's own trajectory scale, forward pass (R = 1) and inverse pass (R = 4,
    noise repeated).  The conditioning features `rhs/b<i>_c` are the trained extractor's (oracle/ref_cpu.py, pinned to the
    reference by tests/golden/forward_*.npz) on `xyz`; they are stored so that a kernel test feeds exactly what ODEfunc saw.
The integrated path has no reference output (torchdiffeq is not installed and not stood in for): the GPU test compares the HIP
path with oracle/cnf_ref.py on these weights - same evaluation / accept / reject counts, and x / z / log-det against the
oracle evaluated in float64 (the anchor that separates kernel error from the fp32 oracle's own).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from modules.continuous.odefunc import ODEfunc, ODEnet          # noqa: E402  (the reference)
from oracle import ref_cpu as O                                    # noqa: E402
from puflow_amd.weights import CNF_COND_CHANNELS, cnf_state_dict_spec, synth_patches   # noqa: E402

SEED = 31


def main():
    torch.set_num_threads(1)
    ck = torch.load("/root/reference/pretrain/puflow-x4-cnf-pu1k.pt", map_location="cpu")
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    sd = {k[len("network."):] if k.startswith("network.") else k: v for k, v in sd.items()}
    want = [k for k, _, _ in cnf_state_dict_spec()]
    assert sorted(want) == sorted(sd.keys()), "checkpoint keys differ from the census"
    arrays = {"meta_checkpoint": np.array("pretrain/puflow-x4-cnf-pu1k.pt"), "meta_torch": np.array(torch.__version__),
              "meta_seed": np.int64(SEED)}
    for k in want:
        arrays["sd/" + k] = sd[k].detach().cpu().numpy()
    B, N = 1, 256
    xyz = synth_patches(B, N, seed=SEED, surface=True)
    g = torch.Generator().manual_seed(SEED)
    noise = torch.stack([torch.randn(B, N, 3, generator=g) for _ in range(6)])
    arrays["xyz"] = xyz.numpy()
    arrays["noise"] = noise.numpy()
    _, idx16 = O.knn_canonical(xyz, xyz, O.K_FEAT)
    cs, _ = O.feat_extract({k: v.float() if v.is_floating_point() else v for k, v in sd.items()}, xyz, idx16)
    for block in range(6):
        cd = CNF_COND_CHANNELS[block]
        f = ODEfunc(ODEnet((64, 64), input_shape=(3,), context_dim=cd, layer_type="concatsquash", nonlinearity="tanh"))
        pfx = f"flow_blocks.{block}.cnf.odefunc."
        f.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}, strict=True)
        T_end = float(sd[f"flow_blocks.{block}.cnf.sqrt_end_time"]) ** 2
        c = cs[block]
        e = noise[block]
        for R in (1, 4):
            t = torch.tensor(T_end * (0.23 + 0.09 * block))
            f.before_odeint_forward(e.clone())
            if R > 1:
                f.before_odeint_inverse(R)
                c_in = torch.repeat_interleave(c, R, dim=1)
            else:
                c_in = c
            # states on the scale the trained flow moves through: the patch itself (block 0, f) ... unit-variance latents
            y = torch.repeat_interleave(xyz, R, dim=1) * (1.0 - block / 8.0) + torch.randn(B, N * R, 3, generator=g) * (0.15 + 0.12 * block)
            logp = torch.zeros(B, N * R, 1)
            dy, ndiv, dc = f(t, (y, logp, c_in))
            assert float(dc.detach().abs().max()) == 0.0
            tag = f"rhs/b{block}_R{R}"
            arrays[f"{tag}_t"] = np.float32(t.item())
            if R == 1:
                arrays[f"rhs/b{block}_c"] = c.detach().numpy().astype(np.float32)      # the conditioning features fed to ODEfunc
            for name, val in (("y", y), ("dy", dy.detach()), ("ndiv", ndiv.detach())):
                arrays[f"{tag}_{name}"] = val.detach().numpy().astype(np.float32)
            print(tag, "T_end %.4f" % T_end, "dy", float(dy.abs().max()), "ndiv", float(ndiv.abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "pretrained_cnf.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
