#!/usr/bin/env python3
"""tests/golden/pretrained_pu1k.npz: the reference's own module with the reference's own TRAINED weights
(`pretrain/puflow-x4-pu1k.pt`), run on CPU on two seeded patches.  Build container only (needs /root/reference, read-only).
The fixture is data: the checkpoint's tensors (an input of the computation - the GPU box has no /root/reference), the seeds of
the input patches and what the reference computed (cs, z, logp, ldj, fz, x).  Harness shims: the same three as
tools/make_golden.py (pytorch3d.ops.knn_points stand-in with the canonical (distance, index) order, np.long, the Gaussian's
default device)."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as G


def main():
    G._install_shims()
    from puflow_amd.weights import synth_patches
    torch.set_num_threads(1)
    ck = torch.load("/root/reference/pretrain/puflow-x4-pu1k.pt", map_location="cpu")
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    sd = {k[len("network."):] if k.startswith("network.") else k: v for k, v in sd.items()}
    net = G.build_reference(sd)
    arrays = {"meta_checkpoint": np.array("pretrain/puflow-x4-pu1k.pt"), "meta_torch": np.array(torch.__version__)}
    for k, v in net.state_dict().items():
        arrays["sd/" + k] = v.detach().cpu().numpy()
    for name, (B, N, seed) in {"a": (2, 256, 11), "b": (1, 2048, 12)}.items():
        xyz = synth_patches(B, N, seed=seed, surface=True)
        out = G.capture(net, xyz)
        arrays[f"{name}/meta"] = np.array([B, N, seed], dtype=np.int64)
        for k in ("cs0", "cs5", "z", "logp", "ldj", "fz", "x", "idx16"):
            a = out[k].numpy()
            arrays[f"{name}/{k}"] = a.astype(np.int16) if k == "idx16" else a
        print(name, "x range", float(out["x"].min()), float(out["x"].max()), "logp", out["logp"].numpy())
    path = os.path.join(ROOT, "tests", "golden", "pretrained_pu1k.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
