"""SyncBN across ranks on the GPU: two gloo ranks that share cuda:0, each with half of the rows, reproduce the
single-process BatchNorm + LeakyReLU on the full batch (outputs, input gradients, parameter gradients, running stats)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4000, 48, generator=g) * 2.0 + 0.7
    w = torch.randn(4000, 48, generator=g)
    gamma = torch.rand(48, generator=g) + 0.5
    beta = torch.randn(48, generator=g) * 0.3
    return x, w, gamma, beta


def _run(x, w, gamma, beta, sync):
    from puflow_amd import train_ops as T
    dev = "cuda:0"
    bn = torch.nn.BatchNorm2d(48).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    xd = x.to(dev).requires_grad_(True)
    with T.sync_bn(sync):
        y = T.bn_lrelu(xd, bn, 0.05)
    loss = (y * w.to(dev)).mean()
    loss.backward()
    return (y.detach().cpu(), xd.grad.cpu(), bn.weight.grad.cpu(), bn.bias.grad.cpu(), bn.running_mean.cpu(), bn.running_var.cpu())


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, w, gamma, beta = _data()
        lo, hi = rank * 2000, (rank + 1) * 2000
        out = _run(x[lo:hi], w[lo:hi], gamma, beta, True)
        q.put((rank,) + tuple(t.numpy() for t in out))
    finally:
        dist.destroy_process_group()


def test_syncbn_two_ranks_equal_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    x, w, gamma, beta = _data()
    y, dx, dg, db, rm, rv = _run(x, w, gamma, beta, False)                  # single process, full batch
    ys = torch.cat([torch.from_numpy(r[1]) for r in res])
    dxs = torch.cat([torch.from_numpy(r[2]) for r in res])
    assert (ys - y).abs().max() < 1e-5
    # each rank back-propagates its LOCAL mean loss: sum over ranks = 2 x the global mean loss
    assert (dxs - 2 * dx).abs().max() < 2e-6 * float(dx.abs().max()) * 2 + 1e-9
    dgs = sum(torch.from_numpy(r[3]) for r in res) / 2
    dbs = sum(torch.from_numpy(r[4]) for r in res) / 2
    assert (dgs - dg).abs().max() < 1e-5 * float(dg.abs().max()) + 1e-8 and (dbs - db).abs().max() < 1e-5 * float(db.abs().max()) + 1e-8
    for r in res:                                                          # both ranks hold the global running statistics
        assert (torch.from_numpy(r[5]) - rm).abs().max() < 1e-6 and (torch.from_numpy(r[6]) - rv).abs().max() < 1e-5
