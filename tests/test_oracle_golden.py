"""The oracle (oracle/ref_cpu.py) against golden vectors captured from the reference itself
(tools/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from puflow_amd.weights import synth_patches, synth_state_dict

CASES = ["n256_s0", "n256_s1", "n2048_s1"]


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"forward_{name}.npz"))
    sd = synth_state_dict(int(g["meta_wseed"]))
    xyz = synth_patches(int(g["meta_B"]), int(g["meta_N"]), seed=int(g["meta_dseed"]), surface=bool(g["meta_surface"]))
    return g, sd, xyz


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(golden_dir, name):
    torch.set_num_threads(1)
    g, sd, xyz = _load(golden_dir, name)
    st = O.forward(sd, xyz, 4, stages=True)
    assert np.array_equal(st["idx16"].numpy(), g["idx16"].astype(np.int64))      # bit-exact indices
    np.testing.assert_allclose(st["x"].numpy(), g["x"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st["z"].numpy(), g["z"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st["fz"].numpy(), g["fz"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(st["ldj"].numpy(), g["ldj"], rtol=1e-6)
    np.testing.assert_allclose(float(st["logp"]), float(g["logp"]), rtol=1e-6)
    for i in range(6):
        n = g[f"cs{i}"].shape[1]
        np.testing.assert_allclose(st["cs"][i].numpy()[:, :n], g[f"cs{i}"], rtol=0, atol=2e-6)


def test_block_trace_and_roundtrip(golden_dir):
    g, sd, xyz = _load(golden_dir, "n256_s0")
    st = O.forward(sd, xyz, 4, stages=True)
    _, _, trace = O.flow_f(sd, xyz, st["cs"])
    for i, (p, ld) in enumerate(trace):
        n = g[f"p{i}"].shape[1]
        np.testing.assert_allclose(p.numpy()[:, :n], g[f"p{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(ld.numpy(), g["block_ld"][i], rtol=1e-6)
    rt = O.flow_block_inverse(sd, 2, O.flow_block_forward(sd, 2, xyz, st["cs"][2])[0], st["cs"][2])
    assert (rt - xyz).abs().max() < 1e-6
    assert float(g["roundtrip_err"]) < 1e-6


def test_knn8_is_prefix_of_knn16():
    xyz = synth_patches(2, 300, seed=11, surface=False)
    xyz[0, 5] = xyz[0, 9]                      # duplicate points -> ties
    _, i16 = O.knn_canonical(xyz, xyz, 16)
    _, i8 = O.knn_canonical(xyz, xyz, 8)
    assert torch.equal(i16[..., :8], i8)
    assert i16[0, 5, 0] == 5 and i16[0, 5, 1] == 9 and i16[0, 9, 0] == 5   # (dist, idx) order on ties


def test_state_dict_census(golden_dir):
    with open(os.path.join(golden_dir, "state_dict_census.json")) as f:
        census = json.load(f)
    sd = synth_state_dict(3)
    assert [k for k, _, _ in census] == list(sd.keys())
    for k, shape, dt in census:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dt
    nparam = sum(v.numel() for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k)
    assert nparam == 806103


def _load_pretrained(golden_dir):
    g = np.load(os.path.join(golden_dir, "pretrained_pu1k.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_matches_reference_with_pretrained_weights(golden_dir, case):
    """The reference's own module with its own TRAINED checkpoint (pretrain/puflow-x4-pu1k.pt, tools/make_golden_pretrained.py):
    the oracle reproduces x, z, the interpolated latent, log-det and log-likelihood on trained-scale activations too."""
    torch.set_num_threads(1)
    g, sd = _load_pretrained(golden_dir)
    B, N, seed = (int(v) for v in g[f"{case}/meta"])
    xyz = synth_patches(B, N, seed=seed, surface=True)
    st = O.forward(sd, xyz, 4, stages=True)
    assert np.array_equal(st["idx16"].numpy(), g[f"{case}/idx16"].astype(np.int64))
    np.testing.assert_allclose(st["x"].numpy(), g[f"{case}/x"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(st["z"].numpy(), g[f"{case}/z"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(st["fz"].numpy(), g[f"{case}/fz"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(st["ldj"].numpy(), g[f"{case}/ldj"], rtol=2e-6)
    np.testing.assert_allclose(float(st["logp"]), float(g[f"{case}/logp"]), rtol=2e-6)
    for i in (0, 5):
        n = g[f"{case}/cs{i}"].shape[-1]
        np.testing.assert_allclose(st["cs"][i].numpy()[..., :n], g[f"{case}/cs{i}"], rtol=0, atol=1e-5)
