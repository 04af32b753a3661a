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


# ---- SyncBN on the FUSED training kernels (csrc/train_fused.hip: a layer's local column sums are all-reduced between its launch
# and a one-block finalisation; train_ops._attach_sync) -------------------------------------------------------------------
def _unit_and_data():
    from puflow_amd.interpflow import _EdgeConvParams
    from puflow_amd.weights import synth_patches
    torch.manual_seed(11)
    p = _EdgeConvParams(32, 64, 16)
    for seq in p.convs:
        seq[1].weight.data.uniform_(0.5, 1.5)
        seq[1].bias.data.uniform_(-0.3, 0.3)
    xyz = synth_patches(4, 256, seed=3)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 256, 32, generator=g)
    w = torch.randn(4, 256, 64, generator=g)
    return p, xyz, x, w


def _mlp_and_data():
    torch.manual_seed(12)
    mlp = torch.nn.Sequential(torch.nn.Conv2d(10, 64, 1), torch.nn.BatchNorm2d(64), torch.nn.LeakyReLU(0.01),
                              torch.nn.Conv2d(64, 64, 1), torch.nn.BatchNorm2d(64), torch.nn.LeakyReLU(0.01), torch.nn.Conv2d(64, 128, 1))
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4096, 10, generator=g)
    w = torch.randn(4096, 128, generator=g)
    return mlp, x, w


def _run_fused(lo, hi, sync):
    """One EdgeConv unit + one BatchNorm MLP on samples / rows [lo, hi) of the fixed data, fused kernels."""
    from puflow_amd import ops, train_ops as T
    dev = "cuda:0"
    p, xyz, x, w = _unit_and_data()
    p = p.to(dev).train()
    xyz, xs, ws = xyz[lo:hi].to(dev), x[lo:hi].to(dev).requires_grad_(True), w[lo:hi].to(dev)
    idx, _ = ops.knn_idx32(xyz, xyz, 16)
    mlp, mx, mw = _mlp_and_data()
    mlp = mlp.to(dev).train()
    r0, r1 = lo * 1024, hi * 1024
    mxs, mws = mx[r0:r1].to(dev).requires_grad_(True), mw[r0:r1].to(dev)
    with T.sync_bn(sync):
        out = T.edgeconv_train_fused(p, xs, idx, True, T.knn_csr(idx), False)
        mo = T.bnmlp_fused(mlp, mxs)
    ((out * ws).sum() + (mo * mws).sum()).backward()                  # sums: a shard's loss is its part of the full-batch loss
    res = {"out": out.detach(), "dx": xs.grad, "mo": mo.detach(), "mdx": mxs.grad}
    for n, q in list(p.named_parameters()) + [("mlp." + n, q) for n, q in mlp.named_parameters()]:
        res["g:" + n] = q.grad
    for i, seq in enumerate(p.convs):
        res[f"rm{i}"], res[f"rv{i}"] = seq[1].running_mean, seq[1].running_var
    res["mrm"], res["mrv"] = mlp[1].running_mean, mlp[1].running_var
    return {k: v.detach().cpu().numpy() for k, v in res.items()}


def _worker_fused(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run_fused(2 * rank, 2 * rank + 2, True)))
    finally:
        dist.destroy_process_group()


def test_fused_syncbn_two_ranks_equal_full_batch():
    """Two ranks with half of the samples each, global-batch statistics on the fused kernels, against ONE process on the whole
    batch: outputs and input gradients row for row, parameter gradients as the sum over the ranks (dgamma / dbeta are local
    sums, like torch.nn.SyncBatchNorm), running statistics equal on both ranks and equal to the full-batch ones."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [r for _, r in sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    full = _run_fused(0, 4, False)

    def close(a, b, what, rtol=2e-5):
        sc = max(float(np.abs(b).max()), 1e-30)
        assert float(np.abs(a - b).max()) / sc < rtol, (what, float(np.abs(a - b).max()) / sc)
    for k in ("out", "dx", "mo", "mdx"):
        close(np.concatenate([r[k] for r in res]), full[k], k)
    for k in full:
        if k.startswith("g:"):
            if k.endswith(".0.bias") or k.endswith("mlp.0.bias") or k.endswith("mlp.3.bias"):
                continue                                                   # a bias in front of BatchNorm: zero gradient, rounding residue
            close(res[0][k] + res[1][k], full[k], k, 2e-4)
        elif k.startswith(("rm", "rv", "mrm", "mrv")):
            close(res[0][k], full[k], k, 1e-5)
            assert np.array_equal(res[0][k], res[1][k]), k               # bit-identical on both ranks: same all-reduced sums
