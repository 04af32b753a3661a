"""Host logic: the algebraic folds of puflow_amd.packing reproduce the oracle (CPU only)."""
import numpy as np
import pytest
import torch

import emulate_plan as E
from oracle import ref_cpu as O
from puflow_amd.packing import fold_state_dict, frag_pack, frag_unpack
from puflow_amd.weights import synth_patches, synth_state_dict


@pytest.mark.parametrize("wseed,dseed,B,N", [(0, 0, 2, 256), (1, 3, 1, 1024)])
def test_folded_plan_matches_oracle(wseed, dseed, B, N):
    sd = synth_state_dict(wseed)
    xyz = synth_patches(B, N, seed=dseed)
    st = O.forward(sd, xyz, 4, stages=True)
    em = E.forward(fold_state_dict(sd), xyz, st["idx16"], stages=True)
    assert (em["x"] - st["x"]).abs().max() < 1e-5            # north_star tolerance on xyz
    assert (em["z"] - st["z"]).abs().max() < 1e-5
    assert ((em["ldj"] - st["ldj"]).abs() / st["ldj"].abs()).max() < 1e-5   # relative (|ldj| ~ 1e3..1e4)
    assert abs(float(em["logp"]) - float(st["logp"])) / abs(float(st["logp"])) < 1e-5
    for i in range(6):
        assert (em["cs"][i] - st["cs"][i]).abs().max() < 1e-5


def test_frag_pack_roundtrip_and_layout():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((35, 50)).astype(np.float32)
    F = frag_pack(W)
    assert F.shape == (3, 4, 64, 4)
    np.testing.assert_array_equal(frag_unpack(F, 35, 50), W)
    # lane l, block (ob,cb), r  <->  W[ob*16 + (l&15)][cb*16 + 4*(l>>4) + r]
    l, ob, cb, r = 37, 1, 2, 3
    assert F[ob, cb, l, r] == W[ob * 16 + (l & 15), cb * 16 + 4 * (l >> 4) + r]


def test_folded_plan_trained_style_weights():
    """Same proof in the wide dynamic range of a trained checkpoint (|h| ~ 300): the folds stay within tolerance."""
    sd = O.calibrate(synth_state_dict(3, style="trained"), synth_patches(4, 256, seed=1))
    xyz = synth_patches(2, 256, seed=2)
    st = O.forward(sd, xyz, 4, stages=True)
    em = E.forward(fold_state_dict(sd), xyz, st["idx16"], stages=True)
    assert (em["x"] - st["x"]).abs().max() < 1e-5
    assert ((em["ldj"] - st["ldj"]).abs() / st["ldj"].abs()).max() < 1e-5
    assert (em["cs"][5] - st["cs"][5]).abs().max() < 1e-5 * float(st["cs"][5].abs().max())
