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
    # PF_EMD_SINGLE: the two ranks SHARE one GPU here; the cooperative EMD auction assumes the process has the device to itself
    # (its workgroups wait for each other: two such kernels at once can starve each other's barriers - csrc/emd.hip).  One
    # workgroup per sample is the documented setting for shared devices; with one GPU per rank it is not needed.
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PF_EMD_SINGLE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from puflow_amd.optim import FusedClipAdam
        from puflow_amd.trainer import TrainerModule, default_cfg
        from puflow_amd.weights import synth_patches, synth_state_dict
        dev = "cuda:0"
        dense = ((synth_patches(4, 1024, seed=11) + 1) / 2)[2 * rank:2 * rank + 2].to(dev)      # this rank's shard
        batch = (dense[:, ::4].contiguous(), dense, torch.ones(2, device=dev))
        tm = TrainerModule(default_cfg(learning_rate=1e-3), loss_mix="pugan")
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
