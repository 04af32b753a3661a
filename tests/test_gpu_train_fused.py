"""GPU: the fused training kernels of one EdgeConv unit (csrc/train_fused.hip) against the un-fused composition
(train_ops.edgeconv_train with PF_TRAIN_FUSED semantics off), which is itself pinned to the oracle's train step in
tests/test_gpu_train.py.  Output, input gradient and every parameter gradient, all unit shapes of the network
(interpflow.py:190-248: growth 8/16/32, pooled K = 16 and the interpolation's un-pooled K = 8 unit with 8 layers)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(cin, odim, growth, seed):
    from puflow_amd.interpflow import _EdgeConvParams
    torch.manual_seed(seed)
    p = _EdgeConvParams(cin, odim, growth)
    for seq in p.convs:                                        # non-trivial BatchNorm parameters
        seq[1].weight.data.uniform_(0.5, 1.5)
        seq[1].bias.data.uniform_(-0.3, 0.3)
    return p.cuda().train()


def _run(p, x, idx, pooling, fused, wout):
    from puflow_amd import train_ops
    for q in p.parameters():
        q.grad = None
    for seq in p.convs:
        seq[1].running_mean.zero_(); seq[1].running_var.fill_(1.0)
    x = x.clone().requires_grad_(True)
    fn = train_ops.edgeconv_train_fused if fused else train_ops.edgeconv_train
    old = train_ops._FUSED
    train_ops._FUSED = fused
    try:
        out = fn(p, x, idx, pooling)
    finally:
        train_ops._FUSED = old
    (out * wout.view_as(out)).sum().backward()
    grads = {n: q.grad.detach().clone() for n, q in p.named_parameters()}
    stats = [(seq[1].running_mean.clone(), seq[1].running_var.clone()) for seq in p.convs]
    return out.detach().clone(), x.grad.detach().clone(), grads, stats


def _close(a, b, what, rtol=2e-4):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err < rtol, f"{what}: max err / max |ref| = {err:.3e}"


def _mask_ambiguous_pools(p, x, idx, wout, K):
    """The two paths round the conv_out outputs differently, so a max-pool whose two best edges agree to ~1e-7 may pick the
    other edge in one of them - the gradient then takes a different, equally valid route and the comparison fails for a
    reason that is not an error.  Pools whose top two candidates are closer than 1e-4 get a zero output gradient."""
    from puflow_amd import train_ops
    with torch.no_grad():
        old = train_ops._FUSED
        train_ops._FUSED = False
        try:
            y = train_ops.edgeconv_train(p, x, idx, pooling=False)
        finally:
            train_ops._FUSED = old
        top = y.view(-1, K, y.shape[-1]).topk(2, dim=1).values
        amb = (top[:, 0] - top[:, 1]) < 1e-4 * (top[:, 0].abs() + 1.0)
        wout = wout.clone()
        wout[amb] = 0.0
    return wout, int(amb.sum())


@pytest.mark.parametrize("cin,odim,growth,K,pooling,B", [
    (3, 32, 8, 16, True, 4), (32, 64, 16, 16, True, 4), (64, 128, 32, 16, True, 4), (128, 128, 32, 16, True, 4),
    (3, 128, 16, 8, False, 4), (128, 128, 32, 16, True, 32),
    (3, 32, 8, 16, False, 4), (32, 64, 16, 16, False, 4), (64, 128, 32, 16, False, 4), (128, 128, 32, 16, False, 4)])
def test_edgeconv_unit_fused_matches_unfused(cin, odim, growth, K, pooling, B):
    from puflow_amd import ops
    from puflow_amd.weights import synth_patches
    N = 256
    xyz = synth_patches(B, N, seed=7).cuda()
    idx16, _ = ops.knn_idx32(xyz, xyz, 16)
    idx = idx16[..., :K].contiguous()
    torch.manual_seed(cin + growth)
    x = xyz if cin == 3 else torch.randn(B, N, cin, device="cuda")
    p = _unit(cin, odim, growth, seed=growth + cin)
    rows = B * N if pooling else B * N * K
    wout = torch.randn(rows, odim, device="cuda")
    if pooling:
        wout, namb = _mask_ambiguous_pools(p, x, idx, wout, K)
        assert namb < 0.01 * wout.numel()
    o_f, dx_f, g_f, st_f = _run(p, x, idx, pooling, True, wout)
    o_u, dx_u, g_u, st_u = _run(p, x, idx, pooling, False, wout)
    _close(o_f, o_u, "output", 2e-5)
    _close(dx_f, dx_u, "dx")
    for n in g_u:
        if n.endswith("0.bias") and "convs" in n:
            # conv bias in front of a BatchNorm: its true gradient is zero, both paths return rounding noise
            assert float(g_f[n].abs().max()) < 1e-2 * float(wout.abs().sum()) * 1e-5 + 1e-3
            continue
        _close(g_f[n], g_u[n], n)
    for (m_f, v_f), (m_u, v_u) in zip(st_f, st_u):
        _close(m_f, m_u, "running_mean", 1e-5)
        _close(v_f, v_u, "running_var", 1e-5)
