"""Chamfer and auction-EMD HIP operators against their CPU oracles (GPU needed)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import emd_ref as E
from oracle import ref_cpu as O
from puflow_amd.weights import synth_patches

DEV = "cuda:0"


def _unit_cube(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, n, 3, generator=g)


@pytest.mark.parametrize("B,N,M", [(2, 256, 1024), (3, 1000, 777), (1, 8192, 8192), (32, 1024, 1024), (40, 300, 500)])
def test_chamfer_forward(B, N, M):
    from puflow_amd import ops
    x = synth_patches(B, N, seed=N, surface=False)
    y = synth_patches(B, M, seed=M + 1, surface=False)
    d1r, i1r, d2r, i2r = O.chamfer_nn(x, y)
    d1, d2, i1, i2 = ops.chamfer_3DDist()(x.to(DEV), y.to(DEV))
    assert torch.equal(d1.cpu(), d1r) and torch.equal(d2.cpu(), d2r)            # bit-exact distances
    assert torch.equal(i1.cpu().long(), i1r) and torch.equal(i2.cpu().long(), i2r)
    loss, _ = ops.chamfer_distance(x.to(DEV), y.to(DEV))
    ref = O.chamfer_distance_mean(x, y)
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))    # north_star: CD within 1e-5
    per = ops.history_chamfer_distance(x.to(DEV), y.to(DEV))
    np.testing.assert_allclose(per.cpu().numpy(), O.chamfer_distance_per_sample(x, y).numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("mode", ["lattice", "duplicates", "far_offset", "tiny", "ragged"])
def test_chamfer_nearest_neighbour_on_hard_clouds(mode):
    """pf_nn1 screens four references at a time with a fused filter distance and computes exact distances only for groups
    that can lower the running minimum: ties (first minimum = lowest index must win), exact duplicates, clouds far from the
    origin, tiny scales and reference counts that are not a multiple of the chunk must match the oracle bit for bit."""
    from puflow_amd import ops
    g = torch.Generator().manual_seed(3)
    B, N, M = 2, 700, 1111 if mode == "ragged" else 1024
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    y = torch.rand(B, M, 3, generator=g) * 2 - 1
    if mode == "lattice":
        x, y = torch.round(x * 4) / 4, torch.round(y * 4) / 4               # 9^3 sites: many equal distances and zeros
    elif mode == "duplicates":
        y[:, 500:600] = y[:, 17:18]                                         # 101 copies of one reference
        x[:, :50] = y[:, 17:18]                                             # queries exactly on it: distance 0, index 17
    elif mode == "far_offset":
        x, y = x * 0.01 + 300.0, y * 0.01 + 300.0
    elif mode == "tiny":
        x, y = x * 1e-18, y * 1e-18                                         # squared distances ~1e-36: near the subnormals
    d1r, i1r, d2r, i2r = O.chamfer_nn(x, y)
    d1, d2, i1, i2 = ops.chamfer_3DDist()(x.to(DEV), y.to(DEV))
    assert torch.equal(d1.cpu(), d1r) and torch.equal(d2.cpu(), d2r)
    assert torch.equal(i1.cpu().long(), i1r) and torch.equal(i2.cpu().long(), i2r)
    # the same clouds as a batch large enough (>= 64 query tiles) for the MFMA-filter kernel (knn5_kernel<1>: pf_nn1's path at
    # the training step's 32 x 1024): every item shifted differently, the first one as above
    reps = 24
    sh = torch.arange(reps).view(reps, 1, 1) * (0.0 if mode in ("tiny", "lattice") else 0.013)
    xb = (x[:1].repeat(reps, 1, 1) + sh).contiguous()
    yb = (y[:1].repeat(reps, 1, 1) + sh).contiguous()
    d1r, i1r, d2r, i2r = O.chamfer_nn(xb, yb)
    d1, d2, i1, i2 = ops.chamfer_3DDist()(xb.to(DEV), yb.to(DEV))
    assert torch.equal(d1.cpu(), d1r) and torch.equal(d2.cpu(), d2r), mode
    assert torch.equal(i1.cpu().long(), i1r) and torch.equal(i2.cpu().long(), i2r), mode


def test_chamfer_backward_matches_autograd():
    from puflow_amd import ops
    x = synth_patches(2, 300, seed=1, surface=False).requires_grad_(True)
    y = synth_patches(2, 200, seed=2, surface=False).requires_grad_(True)
    d = ((x[:, :, None] - y[:, None]) ** 2).sum(-1)
    ref = (d.min(2)[0].mean(1) + d.min(1)[0].mean(1)).mean()
    ref.backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = y.detach().to(DEV).requires_grad_(True)
    loss, _ = ops.chamfer_distance(xd, yd)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-6
    np.testing.assert_allclose(xd.grad.cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(yd.grad.cpu().numpy(), y.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,n,eps,iters", [(2, 256, 0.005, 50), (3, 1024, 0.005, 50), (1, 1024, 0.002, 300),
                                           (2, 192, 0.01, 7), (1, 3072, 0.005, 6), (1, 4608, 0.01, 4)])   # the last two: state (partly) in global memory
def test_emd_matches_oracle(B, n, eps, iters):
    from puflow_amd.loss import emdFunction
    x, y = _unit_cube(B, n, 10 + n), _unit_cube(B, n, 20 + n)
    dref, aref = E.emd_forward(x.numpy(), y.numpy(), eps, iters)
    dist, ass = emdFunction.apply(x.to(DEV), y.to(DEV), eps, iters)
    dist, ass = dist.cpu().numpy(), ass.cpu().numpy()
    # self-consistency (the reference's own check, emd_module.py:91-95): dist == |x - y[assignment]|^2
    ya = np.take_along_axis(y.numpy(), ass[..., None].astype(np.int64), axis=1)
    np.testing.assert_allclose(dist, ((x.numpy() - ya) ** 2).sum(-1), rtol=1e-5, atol=1e-7)
    assert ass.min() >= 0 and ass.max() < n
    # same deterministic rules as the oracle -> same assignment
    assert (ass == aref).mean() > 0.999
    assert abs(dist.sum() - dref.sum()) <= 1e-4 * dref.sum()


def test_emd_quality_and_backward():
    from scipy.optimize import linear_sum_assignment
    from puflow_amd.loss import EarthMoverDistance, emdFunction
    n = 128
    x, y = _unit_cube(1, n, 3), _unit_cube(1, n, 4)
    cost = np.sqrt(((x[0, :, None] - y[0, None]) ** 2).sum(-1).numpy())
    r, c = linear_sum_assignment(cost)
    opt = cost[r, c].sum()
    dist, ass = emdFunction.apply(x.to(DEV), y.to(DEV), 0.0005, 3000)
    got = np.sqrt(dist.cpu().numpy()).sum()
    assert len(np.unique(ass.cpu().numpy())) >= n - 2                  # (almost) a bijection after many iterations
    assert got <= opt + n * 0.0005 * 2 + 1e-3                          # auction bound: within n*eps of optimal
    few = len(np.unique(emdFunction.apply(x.to(DEV), y.to(DEV), 0.005, 3)[1].cpu().numpy()))
    assert few <= len(np.unique(ass.cpu().numpy()))                    # bijection rate rises with iterations
    # backward: 2 g (x - y[assignment]) with the assignment frozen; nothing flows to y
    xd = x.to(DEV).requires_grad_(True)
    yd = y.to(DEV).requires_grad_(True)
    loss = EarthMoverDistance(eps=0.005, iters=50)(xd, yd, radius=torch.tensor([2.0], device=DEV))
    loss.backward()
    d2, a2 = emdFunction.apply(x.to(DEV), y.to(DEV), 0.005, 50)
    gref = E.emd_backward(x.numpy(), y.numpy(), np.full((1, n), 0.5, np.float32), a2.cpu().numpy())
    np.testing.assert_allclose(xd.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-7)
    assert float(yd.grad.abs().sum()) == 0.0
    assert abs(float(loss) - float(d2.sum()) / 2.0) < 1e-4


def test_emd_nan_prediction_row_is_loud_not_a_fault():
    """A NaN prediction (an expected event in training: TrainerModule keeps the reference's NaN-loss guard,
    train_pu1k.py:71-73) must come out as a NaN loss, not as an out-of-bounds bid index inside the auction."""
    from puflow_amd.loss import EarthMoverDistance, emdFunction
    n = 256
    x, y = _unit_cube(2, n, 5), _unit_cube(2, n, 6)
    x[1, 17] = float("nan")
    x[1, 40, 1] = float("inf")
    dist, ass = emdFunction.apply(x.to(DEV), y.to(DEV), 0.005, 50)
    torch.cuda.synchronize()
    ass, dist = ass.cpu().numpy(), dist.cpu().numpy()
    assert ass.min() >= 0 and ass.max() < n                      # every index stays in range
    assert np.isnan(dist[1, 17]) and not np.isfinite(dist[1, 40])
    assert np.isfinite(dist[0]).all()                            # the clean sample is untouched
    dref, aref = E.emd_forward(x.numpy(), y.numpy(), 0.005, 50)
    assert (ass[0] == aref[0]).all() and (ass[1] == aref[1]).mean() > 0.99
    loss = EarthMoverDistance()(x.to(DEV), y.to(DEV))
    assert bool(torch.isnan(loss))


def test_emd_groups_argument_and_status_word():
    """pf_emd_forward_ex: groups = 0 (chosen from the device's occupancy for the kernel), 1 (one workgroup per sample) and a
    cap give the same assignment (same arithmetic and tie rules); a clean run leaves the status word at 0 and
    check_emd_status() silent; a word that was bumped raises once and is cleared."""
    from puflow_amd import _lib
    from puflow_amd.loss import _emd_status, check_emd_status, emdFunction
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 1024, 3, generator=g).cuda()
    y = torch.rand(4, 1024, 3, generator=g).cuda()
    outs = [emdFunction.apply(x, y, 0.005, 50, grp) for grp in (0, 1, 4)]
    for d, a in outs[1:]:
        assert torch.equal(a, outs[0][1]) and torch.equal(d, outs[0][0])
    check_emd_status()                                       # nothing timed out
    st = _emd_status(x.device)
    assert int(st[0]) == 0
    st[0] = 2
    with pytest.raises(_lib.PuflowHipError):
        check_emd_status()
    check_emd_status()                                       # cleared by the report
