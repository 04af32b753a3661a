"""Loss wrappers with the reference's names and call signatures (metric/loss.py:18-42,
metric/emd/emd_module.py:31-79), backed by the HIP library."""
from __future__ import annotations

import os
import torch
import torch.nn as nn
from torch import Tensor
from torch.autograd import Function

from . import _lib, ops


_EMD_STATUS = {}          # device -> int32[2] word the multi-workgroup auction counts timed-out samples in


def _emd_status(dev: torch.device) -> Tensor:
    t = _EMD_STATUS.get(dev)
    if t is None:
        t = _EMD_STATUS[dev] = torch.zeros(2, dtype=torch.int32, device=dev)
    return t


def check_emd_status(device=None) -> None:
    """Raise if a multi-workgroup EMD auction timed out on a grid barrier since the last check (its workgroups were not
    co-resident, e.g. another process on the same GPU held the CUs): the affected samples' distances are NaN.  Reads one
    device word (a synchronisation): called where the host synchronises anyway - never inside a graph capture."""
    for dev, t in list(_EMD_STATUS.items()):
        if device is not None and torch.device(device) != dev:
            continue
        n = int(t[0].item())
        if n:
            t.zero_()
            raise _lib.PuflowHipError(f"pf_emd_forward: {n} workgroup(s) of a multi-workgroup auction timed out on a grid barrier "
                                      "(workgroups not co-resident - is the GPU shared with another process?); their slices "
                                      "of the distances are NaN.  Use EarthMoverDistance(groups=1) on a shared device")


class emdFunction(Function):
    """metric/emd/emd_module.py:31-72.  Unlike the reference there is no n % 1024 / B <= 512 limit.
    groups: workgroups per sample (0 = chosen from the device's occupancy for the kernel, 1 = one workgroup per sample)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters, groups=0):
        lib = _lib.load()
        B, n, _ = xyz1.size()
        assert n == xyz2.size(1) and B == xyz2.size(0)
        xyz1 = ops._f32c(xyz1)
        xyz2 = ops._f32c(xyz2)
        dev = xyz1.device
        # inputs of the auction: price 0, assignment / inverse -1 (emd_module.py:45-56); everything else is scratch the kernel
        # initialises itself - handed over as slices of ONE allocation, in the order (max_inc, bid_inc, max_idx, bid), so that
        # the multi-workgroup kernel can use pairs of them as 64-bit vote words (csrc/emd.hip)
        price = torch.zeros(B, n, device=dev)
        assign2 = torch.full((2, B, n), -1, device=dev, dtype=torch.int32)
        assignment, assignment_inv = assign2[0], assign2[1]
        scratch = torch.empty((5, B, n), device=dev, dtype=torch.int32)
        max_inc, bid_inc, max_idx, bid, unass_idx = scratch[0], scratch[1], scratch[2], scratch[3], scratch[4]
        dist = torch.empty(B, n, device=dev)
        _lib.check(lib.pf_emd_forward_ex(xyz1.data_ptr(), xyz2.data_ptr(), dist.data_ptr(), assignment.data_ptr(),
                                         price.data_ptr(), assignment_inv.data_ptr(), bid.data_ptr(), bid_inc.data_ptr(),
                                         max_inc.data_ptr(), unass_idx.data_ptr(), max_idx.data_ptr(), float(eps),
                                         int(iters), B, n, int(groups), _emd_status(dev).data_ptr(), ops._stream()),
                   "pf_emd_forward")
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        lib = _lib.load()
        xyz1, xyz2, assignment = ctx.saved_tensors
        B, n, _ = xyz1.shape
        graddist = graddist.contiguous().float()
        g1 = torch.zeros_like(xyz1)
        _lib.check(lib.pf_emd_backward(xyz1.data_ptr(), xyz2.data_ptr(), g1.data_ptr(), graddist.data_ptr(),
                                       assignment.data_ptr(), B, n, ops._stream()), "pf_emd_backward")
        return g1, torch.zeros_like(xyz2), None, None, None


class emdModule(nn.Module):
    def forward(self, input1, input2, eps, iters):
        return emdFunction.apply(input1, input2, eps, iters)

    check_status = staticmethod(check_emd_status)


class EarthMoverDistance(nn.Module):
    """metric/loss.py:18-29."""

    def __init__(self, eps=0.005, iters=50, groups=0):
        super().__init__()
        self.eps = eps
        self.iters = iters
        self.groups = groups        # workgroups per sample of the auction: 0 = from the device, 1 = shared-device safe

    check_status = staticmethod(check_emd_status)

    def forward(self, preds, gts, **kwargs):
        loss, _ = emdFunction.apply(preds, gts, self.eps, self.iters, self.groups)
        if kwargs.get("radius") is not None:
            loss = loss / kwargs.get("radius").view(-1, 1)
        return torch.sum(loss)


class ChamferCUDA2(nn.Module):
    """metric/loss.py:32-36 (kaolin form: sum over the batch of per-sample mean+mean)."""

    def forward(self, points1, points2):
        return torch.sum(ops.history_chamfer_distance(points1, points2))


class ChamferCUDA(nn.Module):
    """metric/loss.py:39-42 (pytorch3d form, mean/mean) -> (loss, None)."""

    def forward(self, xyz1: Tensor, xyz2: Tensor, nxyz1: Tensor = None, nxyz2: Tensor = None):
        return ops.chamfer_distance(xyz1, xyz2, x_normals=nxyz1, y_normals=nxyz2, batch_reduction="mean",
                                    point_reduction="mean")


_GRAD_FUSED = os.environ.get("PF_LOSS_GRAD_FUSED", "1") != "0"   # the loss head's backward in two launches (pf_pugan_grad); "0": four


def train_ops_deterministic() -> bool:
    from . import train_ops
    return train_ops.deterministic()


_CD_SIDE = os.environ.get("PF_LOSS_CD_SIDE", "1") != "0"      # Chamfer's nearest neighbours beside the EMD auction (side stream)


class PuganLossFn(Function):
    """The PU-GAN training loss (train_pugan.py:52-67) as ONE autograd node:
        loss = w_logp logp + w_emd sum_b sum_n emd_dist[b, n] / radius[b] + w_cd mean_b chamfer(pred_b, gt_b)
    = emdFunction + _ChamferFn + the torch expressions between them (EarthMoverDistance, ChamferCUDA, the weighted sum), whose
    ~25 one-element launches forward and ~15 backward become one kernel each way (csrc/train_glue.hip); the EMD and Chamfer
    gradient kernels accumulate into one buffer.  apply(pred, gt, radius | None, logp, eps, iters, groups, (w_logp, w_emd, w_cd))
    -> (loss [], terms [3] = weighted EMD, logp, CD for logging; not differentiable)."""

    @staticmethod
    def forward(ctx, pred, gt, radius, logp, eps, iters, groups, weights):
        lib = _lib.load()
        pred, gt = ops._f32c(pred), ops._f32c(gt)
        B, n, _ = pred.shape
        if gt.shape[1] != n:
            raise ValueError("PuganLossFn: prediction and ground truth must have the same number of points (EMD)")
        dev = pred.device
        f32 = dict(dtype=torch.float32, device=dev)
        radius = radius.contiguous().float().view(-1) if radius is not None else None
        logp1 = logp.detach().contiguous().float().view(1)
        # ---- auction (emd_module.py:31-72): price | assignments initialised by one kernel
        price = torch.empty((B, n), **f32)
        assign2 = torch.empty((2, B, n), dtype=torch.int32, device=dev)
        scratch = torch.empty((5, B, n), dtype=torch.int32, device=dev)
        dist = torch.empty((B, n), **f32)
        # ---- Chamfer (metric/loss.py:39-42: mean over points of both directions, mean over the batch) reads pred and gt only, like
        # the auction: on the side stream beside it (two nearest-neighbour launches + two reductions, ~35 us the auction's chain
        # does not wait for).  w_cd = 0: the EMD-only mix of train_pu1k.py:62-67 - no nearest-neighbour search at all
        w_logp, w_emd, w_cd = (float(v) for v in weights)
        per = None
        i1 = i2 = torch.empty((0,), dtype=torch.int32, device=dev)
        cd_side = None
        if w_cd != 0.0 and _CD_SIDE and pred.is_cuda:
            from .train_ops import _side_stream
            cd_side = _side_stream(dev)
            cd_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cd_side):
                d1, d2, i1, i2, per, _ = ops.chamfer_nn(pred, gt)
        _lib.check(lib.pf_emd_init(price.data_ptr(), assign2.data_ptr(), B * n, ops._stream()), "pf_emd_init")
        max_inc, bid_inc, max_idx, bid, unass_idx = scratch[0], scratch[1], scratch[2], scratch[3], scratch[4]
        _lib.check(lib.pf_emd_forward_ex(pred.data_ptr(), gt.data_ptr(), dist.data_ptr(), assign2[0].data_ptr(), price.data_ptr(),
                                         assign2[1].data_ptr(), bid.data_ptr(), bid_inc.data_ptr(), max_inc.data_ptr(),
                                         unass_idx.data_ptr(), max_idx.data_ptr(), float(eps), int(iters), B, n, int(groups),
                                         _emd_status(dev).data_ptr(), ops._stream()), "pf_emd_forward")
        if cd_side is not None:
            torch.cuda.current_stream().wait_stream(cd_side)
            for t in (d1, d2, i1, i2, per):
                t.record_stream(torch.cuda.current_stream())
        elif w_cd != 0.0:
            d1, d2, i1, i2, per, _ = ops.chamfer_nn(pred, gt)
        out = torch.empty((4,), **f32)
        _lib.check(lib.pf_pugan_loss_fwd(logp1.data_ptr(), dist.data_ptr(), radius.data_ptr() if radius is not None else None,
                                         per.data_ptr() if per is not None else None, B, n, w_logp, w_emd, w_cd, out.data_ptr(),
                                         ops._stream()), "pf_pugan_loss_fwd")
        ctx.save_for_backward(pred, gt, assign2, i1, i2, *(() if radius is None else (radius,)))
        ctx.cfg = (B, n, (w_logp, w_emd, w_cd), radius is not None, logp.shape)
        terms = out[1:4]
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)                   # no zero-filled gradient for `terms` (a launch per step)
        return out[0], terms

    @staticmethod
    def backward(ctx, g, _gterms):
        lib = _lib.load()
        B, n, (w_logp, w_emd, w_cd), has_r, lshape = ctx.cfg
        sv = list(ctx.saved_tensors)
        pred, gt, assign2, i1, i2 = sv[:5]
        radius = sv[5] if has_r else None
        dev = pred.device
        f32 = dict(dtype=torch.float32, device=dev)
        if g is None:                                      # (set_materialize_grads(False): only `terms` was used)
            return (None,) * 8
        g1d = g.contiguous().float().view(1)
        if _GRAD_FUSED and not ctx.needs_input_grad[1] and not train_ops_deterministic():
            # the prediction's gradient in two launches: own terms (EMD + first Chamfer direction) stored, the second direction
            # scattered onto them (pf_pugan_grad) - it was four, all on the chain between the auction and the flow's backward
            dlogp = torch.empty((1,), **f32)
            gx = torch.empty_like(pred)
            cd = w_cd != 0.0
            _lib.check(lib.pf_pugan_grad(g1d.data_ptr(), radius.data_ptr() if radius is not None else None, pred.data_ptr(),
                                         gt.data_ptr(), assign2[0].data_ptr(), i1.data_ptr() if cd else None,
                                         i2.data_ptr() if cd else None, B, n, n, w_logp, w_emd, w_cd, gx.data_ptr(), dlogp.data_ptr(),
                                         ops._stream()), "pf_pugan_grad")
            return gx, None, None, dlogp.view(lshape), None, None, None, None
        seeds = torch.empty((3, B, n), **f32)              # graddist | g1 | g2
        dlogp = torch.empty((1,), **f32)
        gx, gy = torch.empty_like(pred), torch.empty_like(gt)
        _lib.check(lib.pf_pugan_loss_bwd(g1d.data_ptr(), radius.data_ptr() if radius is not None else None, B, n, n, w_logp, w_emd,
                                         w_cd, seeds[0].data_ptr(), seeds[1].data_ptr(), seeds[2].data_ptr(), dlogp.data_ptr(),
                                         gx.data_ptr(), gy.data_ptr(), ops._stream()), "pf_pugan_loss_bwd")
        if w_cd != 0.0:
            _lib.check(ops._chamfer_bwd(lib)(pred.data_ptr(), gt.data_ptr(), i1.data_ptr(), i2.data_ptr(), seeds[1].data_ptr(),
                                          seeds[2].data_ptr(), gx.data_ptr(), gy.data_ptr(), B, n, n, ops._stream()), "pf_chamfer_bwd")
        _lib.check(lib.pf_emd_backward(pred.data_ptr(), gt.data_ptr(), gx.data_ptr(), seeds[0].data_ptr(), assign2[0].data_ptr(),
                                       B, n, ops._stream()), "pf_emd_backward")
        # the ground truth's gradient: the Chamfer term's only (the EMD has none, emd_module.py:68-72) - like _ChamferFn returns it
        return gx, (gy if ctx.needs_input_grad[1] else None), None, dlogp.view(lshape), None, None, None, None
