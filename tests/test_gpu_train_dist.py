"""GPU: the multi-rank training step.  Two gloo ranks share cuda:0, each trains on its own half of the batch; after eager
and graph-replayed steps (graph A = forward + backward + one concatenation, eager all-reduce of the flat gradient, graph B =
fused clip + Adam) both ranks must hold bit-identical parameters - the averaged flat gradient is the same buffer content on
both, so any divergence means the bucket / all-reduce / optimizer wiring is wrong."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    # The two ranks SHARE one GPU here.  The EMD auction's occupancy query (csrc/emd.hip pf_emd_forward_ex) knows nothing about
    # another process on the device, so shared-device callers pass groups = 1 (cfg.emd_workgroups=1: one workgroup per sample,
    # no inter-workgroup waits) - as bench.py does under PF_BENCH_SINGLE_DEVICE.  The multi-workgroup auction (groups = 0) has
    # its own test on an unshared device: tests/test_gpu_losses.py::test_emd_groups_argument_and_status_word.
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from puflow_amd.optim import FusedClipAdam
        from puflow_amd.trainer import TrainerModule, default_cfg
        from puflow_amd.weights import synth_patches, synth_state_dict
        dev = "cuda:0"
        dense = ((synth_patches(4, 1024, seed=11) + 1) / 2)[2 * rank:2 * rank + 2].to(dev)      # this rank's shard
        batch = (dense[:, ::4].contiguous(), dense, torch.ones(2, device=dev))
        tm = TrainerModule(default_cfg(learning_rate=1e-3, emd_workgroups=1), loss_mix="pugan")
        tm.network.load_state_dict(synth_state_dict(21))
        tm = tm.to(dev)
        opt = tm.configure_optimizers()["optimizer"]
        assert isinstance(opt, FusedClipAdam)
        losses = [float(tm.train_step(batch, opt)) for _ in range(2)]
        step = tm.graphed_train_step(batch, opt)
        losses += [float(step(batch)) for _ in range(2)]
        flat = torch.cat([p.detach().reshape(-1) for p in tm.parameters()]).cpu().numpy()
        q.put((rank, losses, flat, float(opt.step_t)))
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_keeps_parameters_identical():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, l0, w0, s0), (_, l1, w1, s1) = res
    assert all(np.isfinite(l0)) and all(np.isfinite(l1))
    assert l0 != l1                                     # different shards: different local losses
    assert s0 == s1 and s0 >= 4                         # same number of optimizer updates (the graphed warm-up adds two)
    assert np.array_equal(w0, w1), float(np.abs(w0 - w1).max())
    from puflow_amd.weights import synth_state_dict
    assert np.abs(w0).sum() > 0 and w0.shape[0] == 806103


# ---- first multi-rank step: ActNorm's data-dependent init needs one extra forward; the BatchNorm buffers must not see it
def _bn_buffers(tm):
    return {k: v.detach().cpu().numpy().copy() for k, v in tm.network.named_buffers()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


def _make_tm(dev, sync, emd_workgroups=0):
    from puflow_amd.trainer import TrainerModule, default_cfg
    from puflow_amd.weights import synth_state_dict
    tm = TrainerModule(default_cfg(learning_rate=1e-3, sync_batchnorm=sync, emd_workgroups=emd_workgroups), loss_mix="pugan")
    tm.network.load_state_dict(synth_state_dict(21))
    for b in tm.network.flow_blocks:
        b.actnorm.is_inited = False                       # a fresh model: the first step runs the data-dependent init
    return tm.to(dev)


def _worker_bn(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from puflow_amd.weights import synth_patches
        dev = "cuda:0"
        dense = ((synth_patches(4, 1024, seed=11) + 1) / 2)[2 * rank:2 * rank + 2].to(dev)
        batch = (dense[:, ::4].contiguous(), dense, torch.ones(2, device=dev))
        tm = _make_tm(dev, True, emd_workgroups=1)            # two processes on one device: no grid barriers (see _worker)
        opt = tm.configure_optimizers()["optimizer"]
        loss = float(tm.train_step(batch, opt))
        q.put((rank, loss, _bn_buffers(tm)))
    finally:
        dist.destroy_process_group()


def test_first_multi_rank_step_leaves_batchnorm_buffers_like_one_process():
    """Two ranks with global-batch BatchNorm statistics (cfg.sync_batchnorm) against ONE process on the concatenated batch:
    after the first step - which includes the extra ActNorm-init forward on every rank - running means / variances agree to
    fp32 rounding and the batch counters are equal (the init forward's BatchNorm side effects are undone)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from puflow_amd.weights import synth_patches
    dev = "cuda:0"
    dense = ((synth_patches(4, 1024, seed=11) + 1) / 2).to(dev)
    tm = _make_tm(dev, False)
    opt = tm.configure_optimizers()["optimizer"]
    tm.train_step((dense[:, ::4].contiguous(), dense, torch.ones(4, device=dev)), opt)
    one = _bn_buffers(tm)
    for rank, loss, bufs in res:
        assert np.isfinite(loss)
        assert bufs.keys() == one.keys() and len(one) == 3 * 36            # 36 BatchNorm layers
        for k, v in one.items():
            if k.endswith("num_batches_tracked"):
                assert int(bufs[k]) == int(v), (k, bufs[k], v)              # ONE step counted, not two
            else:
                np.testing.assert_allclose(bufs[k], v, rtol=2e-5, atol=2e-6, err_msg=f"rank {rank} {k}")
