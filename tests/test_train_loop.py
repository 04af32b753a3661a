"""Host logic of the training entry point (puflow_amd/train.py, data.py) - no GPU: the network step is stubbed, what is
tested is what the reference leaves to Lightning or gets wrong (train_pu1k.py:48-51,124-176): the ReduceLROnPlateau is
actually stepped on the logged CD, the warm start, the end-of-run save guard, the batch forms and the rank sharding."""
import os

import numpy as np
import pytest
import torch

from puflow_amd.data import PatchData, SyntheticPatchData, load_patch_arrays
from puflow_amd.train import fit, train
from puflow_amd.trainer import TrainerModule, default_cfg


def _stub(module, cds):
    """Replace the GPU parts of a TrainerModule: constant loss, validation CD taken from `cds` per epoch."""
    it = iter(cds)
    state = {"steps": 0}

    def train_step(batch, optimizer, clip=1e-2):
        state["steps"] += 1
        optimizer.zero_grad(set_to_none=True)
        return torch.tensor(0.1)

    module.train_step = train_step
    module.validation_step = lambda b, i=0: {"vloss": torch.tensor(0.0), "CD": next(it)}
    return state


def test_scheduler_is_stepped_on_cd_and_lr_drops():
    cfg = default_cfg(learning_rate=1e-3, sched_patience=2, sched_factor=0.5)
    m = TrainerModule(cfg)
    st = _stub(m, [1.0] * 20)                               # flat CD: no improvement after the first epoch
    hist = fit(m, train_data=[0, 1, 2], val_data=[0], max_epochs=16, log=None)
    assert st["steps"] == 48 and hist["epochs"] == 16 and m.epoch == 16
    lr = hist["lr"]
    assert lr[0] == 1e-3 and lr[2] == 1e-3                  # within patience
    assert lr[3] == pytest.approx(5e-4)                     # patience 2 exceeded after epochs 1..3 without improvement
    assert lr[6] == pytest.approx(2.5e-4) and lr[-1] >= 1e-4 - 1e-12      # keeps halving, floor min_lr = 1e-4 (train_pu1k.py:50)
    assert min(lr) == pytest.approx(1e-4)
    # an improving CD keeps the rate
    m2 = TrainerModule(cfg)
    _stub(m2, [1.0 / (k + 1) for k in range(12)])
    assert set(fit(m2, [0], [0], 12, log=None)["lr"]) == {1e-3}


def test_train_entry_warm_start_and_save_guard(tmp_path):
    from puflow_amd.interpflow import PointInterpFlow
    src = PointInterpFlow(3)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01 * torch.randn_like(p))
    begin = str(tmp_path / "begin.pt")
    torch.save(src.state_dict(), begin)
    ck = str(tmp_path / "out" / "puflow-test.ckpt")
    m = TrainerModule(default_cfg())
    _stub(m, [1.0] * 50)
    mod, hist = train("Train", ck, begin, train_data=[0], val_data=[0], max_epochs=11, device="cpu", log=None, module=m)
    assert all(b.actnorm.is_inited for b in mod.network.flow_blocks)                 # set_to_initialized_state after the load
    for (k, a), (_, b) in zip(src.state_dict().items(), mod.network.state_dict().items()):
        assert torch.equal(a, b), k                                                   # stubbed steps: weights = warm start
    saved = ck.replace(".ckpt", "-epoch11.ckpt")
    assert os.path.exists(saved)                                                      # > 10 epochs, complete run
    back = torch.load(saved)
    assert list(back.keys()) == list(src.state_dict().keys())
    # short runs are not saved (train_pu1k.py:173), other phases do nothing
    m2 = TrainerModule(default_cfg()); _stub(m2, [1.0] * 50)
    ck2 = str(tmp_path / "short.ckpt")
    train("Train", ck2, None, train_data=[0], val_data=[0], max_epochs=3, device="cpu", log=None, module=m2)
    assert not os.path.exists(ck2.replace(".ckpt", "-epoch3.ckpt"))
    _, h = train("Test", ck2, None, device="cpu", log=None, module=TrainerModule(default_cfg()))
    assert h is None


def test_batch_forms():
    sp, de, r = torch.zeros(1, 4, 16, 3), torch.zeros(1, 4, 64, 3), torch.ones(1, 4)
    a = TrainerModule._unpack({"input_sparse_xyz_pl": sp, "gt_dense_xyz_pl": de, "up_ratio_pl": r})
    assert a[0].shape == (4, 16, 3) and a[1].shape == (4, 64, 3) and a[2].shape == (4,)
    b = TrainerModule._unpack((sp[0], de[0], r[0]))
    assert b[0].shape == (4, 16, 3) and b[2].shape == (4,)
    c = TrainerModule._unpack((sp[0], de[0]))
    assert c[2] is None


def test_patch_file_normalisation_and_sharding(tmp_path):
    rng = np.random.default_rng(0)
    inp = rng.normal(size=(10, 32, 3)).astype(np.float32) * 3 + 5
    gt = np.concatenate([inp, inp + 0.01, inp - 0.01, inp * 1.001], axis=1).astype(np.float32)
    path = str(tmp_path / "patches.npz")
    np.savez(path, poisson_32=inp, poisson_128=gt)
    a, g, rad = load_patch_arrays(path, num_point=32, up_ratio=4)
    assert a.shape == (10, 32, 3) and g.shape == (10, 128, 3) and np.all(rad == 1)
    np.testing.assert_allclose(a.mean(axis=1), 0, atol=1e-5)                          # centred on the input centroid
    np.testing.assert_allclose(np.sqrt((a ** 2).sum(-1)).max(axis=1), 1, atol=1e-6)   # max norm 1 (fetcher.py:34-36)
    np.testing.assert_allclose(g[:, :32], a, atol=1e-6)                               # gt shares the input's centre / scale
    full = PatchData(a, g, rad, batch_size=4, num_point_patch=32, is_augment=True, seed=3)
    parts = [PatchData(a, g, rad, batch_size=4, num_point_patch=32, is_augment=True, seed=3, rank=r, world=2) for r in range(2)]
    assert len(full) == 2
    for bf, b0, b1 in zip(full, parts[0], parts[1]):
        for k in bf:
            assert torch.equal(bf[k], torch.cat([b0[k], b1[k]]))                      # contiguous shards of the same batch
        # augmentation is a similarity transform applied to input and gt alike (up to the input jitter <= 0.03)
        s = bf["up_ratio_pl"]
        assert torch.all((s >= 0.8) & (s <= 1.2))
        d = (bf["gt_dense_xyz_pl"][:, :32] - bf["input_sparse_xyz_pl"]).abs().max()
        assert float(d) <= 0.03 * 1.2 * np.sqrt(3) + 1e-5
    syn = SyntheticPatchData(num_patches=8, num_point_patch=32, up_ratio=4, batch_size=4)
    b = next(iter(syn))
    assert b["input_sparse_xyz_pl"].shape == (4, 32, 3) and b["gt_dense_xyz_pl"].shape == (4, 128, 3)
