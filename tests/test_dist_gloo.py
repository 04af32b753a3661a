"""The N>1 host logic on CPU: world_size-2 gloo processes (sharding, flat-bucket gradient all-reduce,
weight broadcast, output gather, max-over-ranks timing)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from puflow_amd import dist as D
        from puflow_amd.interpflow import PointInterpFlow
        torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
        net = PointInterpFlow(3)
        D.broadcast_module(net, src=0)
        # numpy, not a tensor: a tensor crosses the queue as a shared-memory file that disappears when this process exits
        w = net.merge_convs[2].conv1.weight.detach().clone().numpy()
        # shards cover the batch exactly once
        total = 37
        lo, hi = D.shard_bounds(total, rank, world)
        x = torch.arange(total * 6, dtype=torch.float32).view(total, 2, 3)
        back = D.gather_shards(D.shard_batch(x, rank, world) * 2, total, rank, world)
        # one flat all-reduce averages every gradient
        for i, p in enumerate(net.parameters()):
            p.grad = torch.full_like(p, float(rank + 1)) if i % 2 == 0 else None
        bucket = D.FlatGradBucket(net.parameters())
        bucket.all_reduce_mean()
        g = [float(p.grad.flatten()[0]) for p in list(net.parameters())[:2]]
        tmax = D.max_over_ranks(1.0 + rank, "cpu")
        # BN running statistics drift apart with local batch statistics; rank 0's become authoritative before eval
        bn = net.feat_convs[1].convs[0][1]
        bn.running_mean.fill_(float(rank + 1))
        D.broadcast_buffers(net, src=0)
        rm = float(bn.running_mean[0])
        q.put((rank, lo, hi, bool(torch.equal(back, x * 2)), w, g, bucket.numel, tmax, rm))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, ok0, w0, g0, n0, t0, rm0), (r1, lo1, hi1, ok1, w1, g1, n1, t1, rm1) = res
    assert rm0 == rm1 == 1.0                                     # buffers follow rank 0
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37) and ok0 and ok1
    assert (w0 == w1).all()                                      # broadcast made the weights identical
    assert n0 == n1 == 806103                                    # the whole model is one 3.2 MB bucket
    assert g0 == g1 == [1.5, 0.0]                                # mean of (1, 2); absent grads count as zero
    assert t0 == t1 == 2.0


def test_shard_bounds_cover():
    from puflow_amd.dist import shard_bounds
    for total in (1, 7, 32, 78, 256):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
