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
        # fused == "csr": the neighbour scatter-add of the backward as a gather over the transposed lists
        out = fn(p, x, idx, pooling, train_ops.knn_csr(idx)) if fused == "csr" else fn(p, x, idx, pooling)
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
    o_c, dx_c, g_c, _ = _run(p, x, idx, pooling, "csr", wout)
    _close(dx_c, dx_u, "dx (csr gather)")
    for n in g_u:
        if not (n.endswith("0.bias") and "convs" in n):
            _close(g_c[n], g_u[n], n + " (csr gather)")

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


@pytest.mark.parametrize("kind,cc,td,cdiv,rows", [
    ("merge", 32, 0, 1, 1024), ("merge", 128, 0, 1, 8192), ("cond", 32, 0, 1, 1000 * 1), ("cond", 128, 0, 1, 8192),
    ("cond", 64, 1, 1, 1024), ("cond", 128, 2, 1, 8192), ("cond", 128, 1, 4, 4096), ("cond", 32, 2, 4, 32768)])
def test_mlp_fused_matches_unfused(kind, cc, td, cdiv, rows):
    """csrc/train_mlp.hip against the un-fused composition of the same layers (linear / LeakyReLU kernels + torch cat and
    repeat_interleave): output, dy, dc and every weight / bias gradient."""
    from puflow_amd import train_ops
    from puflow_amd.interpflow import _CondNet, _MergeParams
    torch.manual_seed(cc + td + cdiv)
    T = rows // cdiv
    c0 = torch.randn(T, cc, device="cuda")
    y0 = torch.randn(rows, 3, device="cuda")
    if kind == "merge":
        net = _MergeParams(cc, cc).cuda()
        layers, slopes = [net.conv1, net.conv2], (0.0,)
        dout_w = cc
    else:
        net = _CondNet(td + cc, 64, 3 - td if td else 3).cuda()
        for q in net.parameters():
            if float(q.abs().max()) == 0:
                q.data.normal_(0, 0.1)
        L = net.layers
        layers, slopes = [L[0], L[2], L[4]], (0.01, 0.01)
        dout_w = 3 - td if td else 3
    wout = torch.randn(rows, dout_w, device="cuda")
    # rows with a hidden pre-activation within 1e-5 of the kink get no output gradient: there the two paths may legitimately
    # land on different sides of it (derivative 1 vs slope), which is not an error of either
    with torch.no_grad():
        x = c0.repeat_interleave(cdiv, dim=0) if cdiv > 1 else c0
        if td:
            x = torch.cat([y0[:, :td], x], dim=1)
        z = x @ layers[0].weight.t() + (layers[0].bias if layers[0].bias is not None else 0)
        near = z.abs().min(dim=1).values < 1e-5
        if len(layers) == 3:
            z2 = torch.nn.functional.leaky_relu(z, slopes[0]) @ layers[1].weight.t() + layers[1].bias
            near |= z2.abs().min(dim=1).values < 1e-5
        wout[near] = 0.0
        assert int(near.sum()) < 0.01 * rows

    def run(fused):
        for q in net.parameters():
            q.grad = None
        c = c0.clone().requires_grad_(True)
        y = y0.clone().requires_grad_(True)
        if fused:
            out = train_ops.mlp_fused(y if td else None, c, td, cdiv, slopes, layers)
        else:
            x = c.repeat_interleave(cdiv, dim=0) if cdiv > 1 else c
            if td:
                x = torch.cat([y[:, :td], x], dim=1)
            if kind == "merge":
                out = train_ops.linear(train_ops.ActFn.apply(train_ops.linear(x, net.conv1.weight, net.conv1.bias), 0.0),
                                       net.conv2.weight)
            else:
                out = train_ops.cond_net(net, x)
        (out * wout).sum().backward()
        return (out.detach().clone(), c.grad.clone(), y.grad.clone() if td else None,
                {n: q.grad.clone() for n, q in net.named_parameters()})

    o_f, dc_f, dy_f, g_f = run(True)
    o_u, dc_u, dy_u, g_u = run(False)
    _close(o_f, o_u, "output", 1e-5)
    _close(dc_f, dc_u, "dc", 1e-4)
    if td:
        _close(dy_f, dy_u, "dy", 1e-4)
    for n in g_u:
        _close(g_f[n], g_u[n], n, 1e-4)


@pytest.mark.parametrize("two_inputs,rows", [(False, 4096), (True, 4096), (True, 65536)])
def test_bnmlp_fused_matches_unfused(two_inputs, rows):
    """The interpolation module's BatchNorm MLPs (csrc/train_fused.hip, pf_bnmlp_train_*) against the un-fused layer kernels:
    DistanceEncoder (10 -> 64 -> 64 -> 128, input without gradient) and WeightEstimationUnit (cat[128, 128] -> 128 -> 64 -> 32)."""
    from puflow_amd import train_ops
    from puflow_amd.interpflow import _InterpParams
    torch.manual_seed(5 + rows)
    ip = _InterpParams().cuda().train()
    mlp = ip.weight_unit.mlp if two_inputs else ip.knn_context.distance_encoder.mlp
    for m in mlp:
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    xa0 = torch.randn(rows, 128 if two_inputs else 10, device="cuda")
    xb0 = torch.randn(rows, 128, device="cuda") if two_inputs else None
    wout = torch.randn(rows, 32 if two_inputs else 128, device="cuda")
    # rows with a hidden pre-activation close to the LeakyReLU kink get no output gradient (see test_mlp_fused_matches_unfused)
    with torch.no_grad():
        ref = train_ops._mlp_bn(mlp, torch.cat([xa0, xb0], 1) if two_inputs else xa0)
    def run(fused):
        for q in mlp.parameters():
            q.grad = None
        for m in mlp:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0)
        xa = xa0.clone().requires_grad_(True)
        xb = xb0.clone().requires_grad_(True) if two_inputs else None
        if fused:
            out = train_ops.bnmlp_fused(mlp, xa, xb)
        else:
            out = train_ops._mlp_bn(mlp, torch.cat([xa, xb], 1) if two_inputs else xa)
        (out * wout).sum().backward()
        return (out.detach().clone(), xa.grad.clone(), xb.grad.clone() if two_inputs else None,
                {n: q.grad.clone() for n, q in mlp.named_parameters()},
                [(m.running_mean.clone(), m.running_var.clone()) for m in mlp if isinstance(m, torch.nn.BatchNorm2d)])
    o_f, da_f, db_f, g_f, st_f = run(True)
    o_u, da_u, db_u, g_u, st_u = run(False)
    _close(o_f, o_u, "output", 2e-5)
    _close(da_f, da_u, "dxa", 5e-4)
    if two_inputs:
        _close(db_f, db_u, "dxb", 5e-4)
    for n in g_u:
        if n in ("0.bias", "3.bias"):                 # conv bias in front of a BatchNorm: true gradient zero, rounding noise
            continue
        _close(g_f[n], g_u[n], n, 5e-4)
    for (m_f, v_f), (m_u, v_u) in zip(st_f, st_u):
        _close(m_f, m_u, "running_mean", 1e-5)
        _close(v_f, v_u, "running_var", 1e-5)


def test_fused_clip_adam_matches_torch():
    """csrc/optim.hip against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the same gradients, five steps, with a
    learning-rate change in between (what ReduceLROnPlateau does)."""
    from puflow_amd.optim import FusedClipAdam
    torch.manual_seed(3)
    shapes = [(64, 129), (64,), (3, 3), (1, 1, 3), (128, 384, 1, 1), (5000,), (4097,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fa = FusedClipAdam(pa, lr=1e-3, max_norm=1e-2)
    tb = torch.optim.Adam(pb, lr=1e-3)
    for it in range(5):
        grads = [torch.randn_like(p) * (10.0 if it % 2 == 0 else 1e-4) for p in pa]     # clipped and un-clipped steps
        if it == 3:
            fa.param_groups[0]["lr"] = 5e-4
            tb.param_groups[0]["lr"] = 5e-4
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.clone(), g.clone()
        fa.step()
        norm = torch.nn.utils.clip_grad_norm_(pb, 1e-2)
        tb.step()
        assert abs(float(fa.coef[1]) - float(norm)) <= 1e-5 * float(norm)
        for p, q in zip(pa, pb):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), (it, p.shape)
    sd = fa.state_dict()
    fa.load_state_dict(sd)
    assert float(fa.step_t) == 5.0


def test_fused_clip_adam_through_a_gradient_table():
    """pf_clip_adam_ptrs (FusedClipAdam.step_table): the gradients read where autograd left them, through a device table of their
    addresses - what the captured training step uses instead of concatenating ~240 tensors first.  Same update, bit for bit, as
    the flat-buffer entry point on the same gradients (one of them missing: a table entry that points at zeros); the clipped
    gradients are written back into the tensors."""
    from puflow_amd.optim import FusedClipAdam
    torch.manual_seed(7)
    shapes = [(64, 129), (64,), (3, 3), (128, 384, 1, 1), (5000,), (4097,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fa, fb = FusedClipAdam(pa, lr=1e-3, max_norm=1e-2), FusedClipAdam(pb, lr=1e-3, max_norm=1e-2)
    zeros = torch.zeros(max(p.numel() for p in pa), device="cuda")
    for it in range(3):
        grads = [torch.randn_like(p) * (10.0 if it != 1 else 1e-4) for p in pa]
        for k, (p, q, g) in enumerate(zip(pa, pb, grads)):
            p.grad, q.grad = (None, None) if k == 2 else (g.clone(), g.clone())
        ptrs = fa.grad_table_of(zeros)
        assert ptrs is not None and ptrs[2] == zeros.data_ptr()
        fa.step_table(torch.tensor(ptrs, dtype=torch.int64, device="cuda"))
        fb.step()                                                       # concatenation + pf_clip_adam
        assert torch.equal(fa.coef[:3], fb.coef[:3])
        for k, (p, q) in enumerate(zip(pa, pb)):
            assert torch.equal(p, q), (it, k)
            if k != 2:
                assert torch.equal(p.grad, grads[k] * fa.coef[0]), (it, k)          # clipped in place
        assert torch.equal(fa.exp_avg, fb.exp_avg) and torch.equal(fa.exp_avg_sq, fb.exp_avg_sq)
    assert float(zeros.abs().max()) == 0.0
    pa[0].grad = pa[0].grad.t().contiguous().t() if pa[0].grad.dim() == 2 else pa[0].grad      # not contiguous: no table
    assert fa.grad_table_of(zeros) is None


def test_fused_clip_adam_skips_a_non_finite_gradient():
    """A NaN / inf anywhere in the flat gradient (what a timed-out EMD barrier or a NaN loss leaves) must not reach the
    parameters or the moments: the update is skipped ON THE DEVICE (a captured step reads its status words only every few
    replays), counted, and the next finite gradient updates exactly as if the bad step had never happened."""
    from puflow_amd.optim import FusedClipAdam
    torch.manual_seed(5)
    shapes = [(64, 33), (64,), (5000,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fa, fb = FusedClipAdam(pa, lr=1e-3, max_norm=1e-2), FusedClipAdam(pb, lr=1e-3, max_norm=1e-2)
    g1 = [torch.randn_like(p) for p in pa]
    g2 = [torch.randn_like(p) * 1e-4 for p in pa]
    for opt, ps in ((fa, pa), (fb, pb)):
        for p, g in zip(ps, g1):
            p.grad = g.clone()
        opt.step()
    for bad in (float("nan"), float("inf")):
        before = [p.detach().clone() for p in pa]
        m0, v0, s0 = fa.exp_avg.clone(), fa.exp_avg_sq.clone(), float(fa.step_t)
        for p, g in zip(pa, g2):
            p.grad = g.clone()
        pa[2].grad[1234] = bad
        fa.step()
        assert float(fa.coef[2]) == 1.0 and float(fa.step_t) == s0
        assert all(torch.equal(p, b) for p, b in zip(pa, before)) and torch.equal(fa.exp_avg, m0) and torch.equal(fa.exp_avg_sq, v0)
    assert fa.skipped_updates() == 2 and fa.skipped_updates() == 0
    for opt, ps in ((fa, pa), (fb, pb)):
        for p, g in zip(ps, g2):
            p.grad = g.clone()
        opt.step()
    assert float(fa.coef[2]) == 0.0 and float(fa.step_t) == 2.0
    assert all(torch.equal(p, q) for p, q in zip(pa, pb))


def test_training_forward_backward_on_an_odd_patch_size():
    """A patch size whose edge count is not a multiple of 16 (N = 255, K = 8 in the interpolation unit -> 2040 edges): the fused
    EdgeConv unit declines it and the per-op path takes over inside the same step; everything else stays fused; the result
    matches the all-per-op path."""
    from puflow_amd import train_ops
    from puflow_amd.interpflow import PointInterpFlow
    from puflow_amd.weights import synth_patches, synth_state_dict
    xyz = synth_patches(1, 255, seed=3).cuda()

    def run(fused):
        torch.manual_seed(0)
        net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(5)); net = net.cuda().train()
        old = train_ops._FUSED
        train_ops._FUSED = fused
        try:
            u, logp = train_ops.forward_train(net, xyz, 4)
            (u.square().mean() + 1e-4 * logp).backward()
        finally:
            train_ops._FUSED = old
        return u.detach(), float(logp), {n: q.grad.clone() for n, q in net.named_parameters() if q.grad is not None}

    u_f, l_f, g_f = run(True)
    u_u, l_u, g_u = run(False)
    assert (u_f - u_u).abs().max() < 2e-5 and abs(l_f - l_u) < 1e-5 * abs(l_u)
    # parameters whose true gradient is zero - a bias in front of a BatchNorm (also across a unit boundary: the interpolation
    # unit's conv_out bias feeds the weight unit's first BatchNorm) or in front of the softmax over the neighbours - come out as
    # rounding noise in both paths: only gradients above 1e-5 of the largest one are compared
    gmax = max(float(g.abs().max()) for g in g_u.values())
    errs = {n: float((g_f[n] - g_u[n]).abs().max()) / float(g_u[n].abs().max()) for n in g_u
            if float(g_u[n].abs().max()) > 1e-5 * gmax}
    assert len(errs) > 0.8 * len(g_u)
    # no masking of ambiguous max-pools / LeakyReLU kinks here (whole network, 255 points): a flipped pool moves a handful of
    # gradients by a few per cent - hence a distribution criterion instead of a bound on the worst one
    v = np.sort(np.array(list(errs.values())))
    worst = max(errs, key=errs.get)
    assert np.median(v) < 1e-3 and v[int(0.9 * len(v))] < 2e-2 and v[-1] < 0.5, (float(np.median(v)), float(v[int(0.9 * len(v))]), worst, errs[worst])


@pytest.mark.parametrize("R", [2, 3])
def test_training_forward_backward_other_up_ratios(R):
    """Up-ratios other than 4 in train mode (the reference's interpolation supports any R <= r_max): a power of two shares the
    conditioning row in the kernel, any other ratio goes through a materialised repeat - fused path against per-op path."""
    from puflow_amd import train_ops
    from puflow_amd.interpflow import PointInterpFlow
    from puflow_amd.weights import synth_patches, synth_state_dict
    xyz = synth_patches(2, 128, seed=4).cuda()

    def run(fused):
        net = PointInterpFlow(3); net.load_state_dict(synth_state_dict(6)); net = net.cuda().train()
        old = train_ops._FUSED
        train_ops._FUSED = fused
        try:
            u, logp = train_ops.forward_train(net, xyz, R)
            (u.square().mean() + 1e-4 * logp).backward()
        finally:
            train_ops._FUSED = old
        g = net.flow_blocks[3].coupling1.bias_net.layers[2].weight.grad.clone()
        return u.detach(), float(logp), g

    u_f, l_f, g_f = run(True)
    u_u, l_u, g_u = run(False)
    assert u_f.shape == (2, 128 * R, 3)
    assert (u_f - u_u).abs().max() < 2e-5 and abs(l_f - l_u) < 1e-5 * abs(l_u)
    assert float((g_f - g_u).abs().max()) < 2e-2 * float(g_u.abs().max())


def test_fused_batchnorm_statistics_with_a_large_mean():
    """The fused kernels accumulate a layer's batch statistics centred on its running mean (sum (y - p), sum (y - p)^2).  A
    BatchNorm input whose mean is 300 standard deviations away from zero - a conv bias of 30 on activations of std 0.1 - loses
    ~1e-2 of its variance with E[y^2] - E[y]^2 on fp32 partial sums; with running statistics that have tracked the batch (as in
    training after a few steps) the fused path matches the two-pass un-fused kernels and float64."""
    from puflow_amd import train_ops
    from puflow_amd.interpflow import _InterpParams
    torch.manual_seed(3)
    rows = 32768
    mlp = _InterpParams().cuda().train().knn_context.distance_encoder.mlp
    x = torch.randn(rows, 10, device="cuda")
    with torch.no_grad():
        mlp[0].weight.mul_(0.03)
        mlp[0].bias.fill_(30.0)
        y0 = torch.nn.functional.linear(x.double(), mlp[0].weight.view(64, 10).double(), mlp[0].bias.double())
        mean64, var64 = y0.mean(0), y0.var(0, unbiased=False)
        assert float((mean64.abs() / var64.sqrt()).min()) > 100          # the regime this test is about
    outs = {}
    for fused in (True, False):
        with torch.no_grad():
            mlp[1].running_mean.copy_(mean64.float()); mlp[1].running_var.fill_(1.0)     # running statistics that have tracked the data
            mlp[4].running_mean.zero_(); mlp[4].running_var.fill_(1.0)
        out = train_ops.bnmlp_fused(mlp, x) if fused else train_ops._mlp_bn(mlp, x)
        outs[fused] = (out.detach().clone(), mlp[1].running_var.clone())
    (o_f, v_f), (o_u, v_u) = outs[True], outs[False]
    want = 0.9 * 1.0 + 0.1 * var64 * rows / (rows - 1)                                   # momentum 0.1, unbiased variance
    assert float(((v_f.double() - want).abs() / want).max()) < 1e-5
    assert float(((v_u.double() - want).abs() / want).max()) < 1e-5
    assert float((o_f - o_u).abs().max()) < 2e-4 * float(o_u.abs().max())


@pytest.mark.parametrize("cin,odim,growth,B,N", [
    (3, 32, 8, 32, 256), (32, 64, 16, 32, 256), (128, 128, 32, 32, 256), (64, 128, 32, 4, 256),
    (128, 128, 32, 3, 100),                      # 300 tiles: waves without a tile, workgroups with a partial round
    (3, 32, 8, 1, 64), (32, 64, 16, 7, 256)])
def test_edgeconv_unit_persistent_forward_matches_the_per_layer_kernels(cin, odim, growth, B, N):
    """The unit's forward as ONE persistent launch with a grid barrier per BatchNorm layer (ec_fwdp_kernel) against the
    per-layer kernels: the stored pre-BatchNorm tensor is bit-identical (same products in the same order), the statistics
    agree to summation order, the pooled output to the split-fp16 conv_out's operand order, and the backward - which reads what
    the forward stored (Y, aff, argmax) - gives the same gradients.  The barrier words are left zero, the status word clean."""
    from puflow_amd import ops, train_ops
    from puflow_amd.weights import synth_patches
    xyz = synth_patches(B, N, seed=9).cuda()
    idx, _ = ops.knn_idx32(xyz, xyz, 16)
    torch.manual_seed(cin + growth + B)
    x = (xyz if cin == 3 else torch.randn(B, N, cin, device="cuda"))
    p = _unit(cin, odim, growth, seed=B)
    wout = torch.randn(B * N, odim, device="cuda")
    wout, _ = _mask_ambiguous_pools(p, x, idx, wout, 16)
    csr = train_ops.knn_csr(idx)

    def run(persistent):
        for q in p.parameters():
            q.grad = None
        for seq, rm in zip(p.convs, rmeans):                           # non-zero running means: they are the pivots of the statistics
            seq[1].running_mean.copy_(rm)
            seq[1].running_var.fill_(1.0)
        xx = x.clone().requires_grad_(True)
        out = train_ops.edgeconv_train_fused(p, xx, idx, True, csr, persistent)
        saved = out.grad_fn.saved_tensors                              # x, idx, Wpq, PQ, Y, aff, arg, ...
        Y, aff, arg = saved[4].clone(), saved[5].clone(), saved[6].clone()
        (out * wout.view_as(out)).sum().backward()
        grads = {n: q.grad.detach().clone() for n, q in p.named_parameters()}
        stats = [(seq[1].running_mean.clone(), seq[1].running_var.clone()) for seq in p.convs]
        return out.detach().clone(), xx.grad.detach().clone(), grads, stats, Y, aff, arg

    # running means as a trained model has them: near the batch means (0.97 of them), not at zero - every workgroup reads its
    # pivots before the barrier, workgroup 0 updates them after it
    for seq in p.convs:
        seq[1].running_mean.zero_()
    with torch.no_grad():
        train_ops.edgeconv_train_fused(p, x, idx, True, csr, False)
    rmeans = [seq[1].running_mean.clone() * (0.97 / seq[1].momentum) for seq in p.convs]
    assert train_ops._PERSIST
    o_p, dx_p, g_p, st_p, Y_p, aff_p, arg_p = run(True)
    sync = train_ops._sync_words(xyz.device)
    assert sync.tolist() == [0, 0, 0, 0]                               # barrier words back to zero, no timeout
    o_l, dx_l, g_l, st_l, Y_l, aff_l, arg_l = run(False)
    g = growth
    assert torch.equal(Y_p[:, :g], Y_l[:, :g])                         # layer 0: P[i] + Q[j], no statistics involved yet
    _close(Y_p, Y_l, "Y", 2e-6)
    for rw, nm in enumerate(("scale", "shift", "mean", "rstd")):
        _close(aff_p[rw], aff_l[rw], "aff " + nm, 1e-5)
    _close(o_p, o_l, "out", 2e-6)
    same = (arg_p == arg_l) | (wout.view(-1, odim) == 0)
    assert bool(same.all()), int((~same).sum())
    _close(dx_p, dx_l, "dx", 2e-5)
    for n in g_l:
        if ".0.bias" in n:                                             # a conv bias in front of BatchNorm: zero gradient, rounding residue
            continue
        _close(g_p[n], g_l[n], n, 2e-4)
    for (m_p, v_p), (m_l, v_l) in zip(st_p, st_l):
        _close(m_p, m_l, "running_mean", 1e-5)
        _close(v_p, v_l, "running_var", 1e-5)
    train_ops.check_persist_status()


# ---- round 5: pf_gemm on conflict-free LDS images (gemm2_kernel) = the round-1 kernel bit for bit ---------------------------
# operand orientations of the three point GEMMs of an EdgeConv unit (train_fused.hip: PQ = x Wpq^T + b, dx = dPQ Wpq,
# dWpq = dPQ^T x with split-K) plus ragged shapes, every tile shape of gemm_shape()
@pytest.mark.parametrize("M,N,K,a_kfast,b_nfast,bias", [
    (8192, 512, 128, True, False, True),        # PQ forward, 128-channel unit
    (8192, 128, 512, True, True, False),        # dx
    (512, 128, 8192, False, True, False),       # dWpq (split-K slabs + reduce)
    (8192, 128, 64, True, False, True),         # unit 1
    (8192, 64, 32, True, True, False),
    (8192, 16, 128, True, False, True),         # skinny N
    (8192, 32, 64, True, True, False),
    (16, 256, 8192, False, True, False),        # skinny M, split-K
    (64, 128, 8192, False, False, False),
    (1000, 72, 100, True, False, True),         # ragged rows / columns, K not a multiple of 32
    (260, 260, 36, False, True, True),
    (4096, 512, 512, True, False, False),       # 128 x 128 tiles
])
def test_gemm_conflict_free_kernel_is_bit_identical(M, N, K, a_kfast, b_nfast, bias):
    from puflow_amd import train_ops as T
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((M, K) if a_kfast else (K, M), generator=g).cuda()
    Bm = torch.randn((K, N) if b_nfast else (N, K), generator=g).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    sam, sak = (K, 1) if a_kfast else (1, M)
    sbk, sbn = (N, 1) if b_nfast else (1, K)
    out = []
    for arith in (0, 1):
        C = torch.full((M, N), float("nan"), device="cuda")
        T._gemm(A, sam, sak, Bm, sbk, sbn, C, N, b, M, N, K, arith)
        out.append(C)
    torch.cuda.synchronize()
    ref = (A if a_kfast else A.t()).double() @ (Bm if b_nfast else Bm.t()).double()
    if bias:
        ref = ref + b.double()
    assert torch.equal(out[0], out[1]), float((out[0] - out[1]).abs().max())
    assert float((out[0].double() - ref).abs().max()) <= 2e-6 * K ** 0.5 * float(ref.abs().max() + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(4, 256), (3, 100)])
def test_paired_transposed_lists_equal_the_single_builds(B, N):
    """pf_knn_csr_pair: the transposed neighbour lists of idx [B, N, 16] and of its first 8 columns from ONE pass (4 launches) -
    the same offsets and, list by list, the same edge ids as pf_knn_csr on each tensor alone (slots inside a list are handed
    out in arrival order: compared as sorted lists); every edge id appears exactly once."""
    from puflow_amd import ops, train_ops
    from puflow_amd.weights import synth_patches
    xyz = synth_patches(B, N, seed=19).cuda()
    idx16, _ = ops.knn_idx32(xyz, xyz, 16)
    idx8 = idx16[..., :8].contiguous()
    (o16, e16), (o8, e8) = train_ops.knn_csr_pair(idx16, 8)
    for (off, edge), ref_idx, K in (((o16, e16), idx16, 16), ((o8, e8), idx8, 8)):
        roff, redge = train_ops.knn_csr(ref_idx)
        assert torch.equal(off, roff)
        assert torch.equal(torch.sort(edge).values, torch.arange(B * N * K, dtype=torch.int32, device="cuda"))
        seg = torch.repeat_interleave(torch.arange(B * N, device="cuda"), (off[1:] - off[:-1]).long())
        key = seg * (B * N * K) + edge.long()                         # (list, edge id): sorting it sorts every list in place
        rkey = seg * (B * N * K) + redge.long()
        assert torch.equal(torch.sort(key).values, torch.sort(rkey).values)
        tgt = (torch.arange(B * N, device="cuda") // N * N).repeat_interleave(K) + ref_idx.reshape(-1).long()
        assert torch.equal(tgt[edge.long()], seg)                      # every edge sits in the list of the point it points at
