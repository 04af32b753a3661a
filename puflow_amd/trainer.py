"""TrainerModule: the reference's LightningModule surface (modules/discrete/train_pu1k.py:29-105,
train_pugan.py:29-105, train_pugeo.py) without requiring pytorch_lightning (not installed here):
`forward`, `configure_optimizers`, `training_step`, `validation_step`, `validation_epoch_end`, cfg fields
`learning_rate, sched_patience, sched_factor` - plus `train_step()`, the explicit
forward -> loss -> backward -> (RCCL all-reduce) -> clip -> Adam sequence Lightning would run
(gradient_clip_val=1e-2, train_pu1k.py:149).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .dist import FlatGradBucket
from .interpflow import PointInterpFlow
from .loss import ChamferCUDA, ChamferCUDA2, EarthMoverDistance

try:                                     # optional: behave as a LightningModule when Lightning exists
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                        # pragma: no cover - not installed in this image
    _Base = nn.Module


def default_cfg(**kw):
    # sync_batchnorm: BatchNorm statistics over ALL ranks (an extension: the reference trains on one GPU, where it is
    # the same thing; default False = each rank normalises with its own shard, like DDP without SyncBatchNorm)
    # deterministic: bit-reproducible training steps (a debugging switch: BatchNorm statistics as exact fixed-point sums, the
    # remaining float-atomic gradients as ordered gathers; slower - no persistent kernels; train_ops.set_deterministic)
    # persistent_kernels: the training step's grid-barrier kernels (EdgeConv units as one launch each).  None = on, unless
    # emd_workgroups == 1 says the device is shared with other processes (grid barriers need every workgroup resident)
    cfg = dict(net="UpsamplingFlow", learning_rate=1e-3, sched_patience=10, sched_factor=0.5, seed=2021, sync_batchnorm=False,
               fused_optimizer=True, emd_workgroups=0, persistent_kernels=None, deterministic=False, dw_stream=False)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


class TrainerModule(_Base):
    """loss_mix: 'pu1k' / 'pugeo' = 1e-4 logp + 5e-2 EMD (train_pu1k.py:65); 'pugan' adds 1e-1 CD and divides the
    EMD by the patch radius (train_pugan.py:59-61)."""

    def __init__(self, cfg=None, loss_mix: str = "pu1k"):
        super().__init__()
        self.cfg = cfg or default_cfg()
        self.loss_mix = loss_mix
        self.network = PointInterpFlow(pc_channel=3)
        self.network.sync_batchnorm = bool(getattr(self.cfg, "sync_batchnorm", False))     # an argument of its train-mode forward
        self.network.deterministic = bool(getattr(self.cfg, "deterministic", False))
        # weight gradients on their own stream beside the backward chain (train_ops._dw_begin): measured slower at the bench shape
        # (4.54 -> 4.79 ms per step), so an opt-in
        self.network.train_dw_stream = bool(getattr(self.cfg, "dw_stream", False))
        pk = getattr(self.cfg, "persistent_kernels", None)
        self.network.train_persistent = bool(pk) if pk is not None else int(getattr(self.cfg, "emd_workgroups", 0)) != 1
        # cfg.emd_workgroups: workgroups per sample of the EMD auction (0 = chosen from the device, 1 = safe on a GPU that is
        # shared with other processes: no inter-workgroup waits; csrc/emd.hip)
        self.emd_loss = EarthMoverDistance(groups=int(getattr(self.cfg, "emd_workgroups", 0)))
        self._nan_subs: Optional[Tensor] = None       # device counter: NaN losses replaced inside captured steps
        self.chamfer_loss = ChamferCUDA()
        self.chamfer_loss2 = ChamferCUDA2()
        self.epoch = 0
        self.min_CD = 100.0
        self.min_Nor = 15.0
        self.logged = {}
        self._bucket: Optional[FlatGradBucket] = None

    def log(self, name, value, **_kw):          # Lightning's self.log when running without Lightning
        if _Base is nn.Module:
            # kept as a (detached) device tensor: no device->host sync per logged value, readable inside a captured step;
            # `logged_values()` converts on demand
            self.logged[name] = value.detach() if isinstance(value, Tensor) else float(value)
        else:                                   # pragma: no cover
            super().log(name, value, **_kw)

    def logged_values(self) -> dict:
        return {k: float(v) for k, v in self.logged.items()}

    def forward(self, p: Tensor, **kwargs):
        return self.network(p, **kwargs)

    def configure_optimizers(self):
        # Adam as in the reference (train_pu1k.py:46).  On the GPU the update and Lightning's gradient clipping run as two fused
        # launches (puflow_amd/optim.py: same arithmetic as clip_grad_norm_ + torch.optim.Adam; cfg.fused_optimizer=False keeps
        # torch.optim.Adam)
        on_gpu = next(self.parameters()).is_cuda
        if on_gpu and getattr(self.cfg, "fused_optimizer", True):
            from .optim import FusedClipAdam
            optimizer = FusedClipAdam(self.parameters(), lr=self.cfg.learning_rate, max_norm=1e-2)
        else:
            optimizer = torch.optim.Adam(self.parameters(), lr=self.cfg.learning_rate)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, "min", factor=self.cfg.sched_factor, min_lr=1e-4,
                                                               patience=self.cfg.sched_patience)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "monitor": "CD"}}

    @staticmethod
    def _unpack(batch):
        """The reference's two batch forms: the PU1K / PUGeo dict with a leading DataLoader dimension of 1
        (train_pu1k.py:55-56: 'gt_dense_xyz_pl', 'input_sparse_xyz_pl', squeezed) and the PU-GAN tuple
        (xyz_sparse, xyz_dense[, radius]) (train_pugan.py:55)."""
        if isinstance(batch, dict):
            xyz_sparse, xyz_dense = batch["input_sparse_xyz_pl"], batch["gt_dense_xyz_pl"]
            radius = batch.get("up_ratio_pl", batch.get("pointclouds_radius"))
        elif len(batch) == 3:
            xyz_sparse, xyz_dense, radius = batch
        else:
            (xyz_sparse, xyz_dense), radius = batch, None
        if xyz_sparse.dim() == 4:
            xyz_sparse, xyz_dense = xyz_sparse.squeeze(0), xyz_dense.squeeze(0)
            radius = radius.squeeze(0) if radius is not None and radius.dim() == 2 else radius
        return xyz_sparse, xyz_dense, radius

    def _losses(self, batch):
        xyz_sparse, xyz_dense, radius = self._unpack(batch)
        upratio = int(xyz_dense.shape[1] / xyz_sparse.shape[1])
        xyz_pred, logpx = self(xyz_sparse, upratio=upratio)
        if self.loss_mix != "pugan" and self._fused_loss(xyz_pred, xyz_dense):
            # train_pu1k.py:62-67: logp + EMD (no radius, no Chamfer term) through the same node
            from .loss import PuganLossFn
            loss, terms = PuganLossFn.apply(xyz_pred, xyz_dense, None, logpx, self.emd_loss.eps, self.emd_loss.iters,
                                            self.emd_loss.groups, (1e-4, 5e-2, 0.0))
            self.log("EMD", terms[0])
            self.log("logpx", terms[1])
            return loss
        if self.loss_mix == "pugan" and self._fused_loss(xyz_pred, xyz_dense):
            # the same three terms from one autograd node (loss.PuganLossFn: the ~40 one-element launches between the EMD /
            # Chamfer kernels and the scalar loss fused into two)
            from .loss import PuganLossFn
            loss, terms = PuganLossFn.apply(xyz_pred, xyz_dense, radius, logpx, self.emd_loss.eps, self.emd_loss.iters,
                                            self.emd_loss.groups, (1e-4, 5e-2, 1e-1))
            self.log("CD", terms[2])
            self.log("EMD", terms[0])
            self.log("logpx", terms[1])
            return loss
        if self.loss_mix == "pugan":
            emd = self.emd_loss(xyz_pred, xyz_dense, radius=radius)
            cd, _ = self.chamfer_loss(xyz_pred, xyz_dense)
            loss = logpx * 1e-4 + emd * 5e-2 + cd * 1e-1
            self.log("CD", cd * 1e-1)
        else:
            emd = self.emd_loss(xyz_pred, xyz_dense)
            loss = logpx * 1e-4 + emd * 5e-2
        self.log("EMD", emd * 5e-2)
        self.log("logpx", logpx * 1e-4)
        return loss

    @staticmethod
    def _fused_loss(pred, gt) -> bool:
        from . import train_ops
        return train_ops._GLUE and pred.is_cuda and pred.shape[1] == gt.shape[1]

    def training_step(self, batch, batch_idx=0):
        loss = self._losses(batch)
        if self.loss_mix != "pugan":                                  # NaN guard, train_pu1k.py:71-73
            if torch.cuda.is_current_stream_capturing():
                # inside a captured step nothing may read the device: the same substitution as a tensor op (a NaN loss
                # becomes the constant 0.1, whose gradient is zero instead of the reference's NaN gradients)
                bad = torch.isnan(loss)
                if self._nan_subs is None or self._nan_subs.device != loss.device:
                    raise RuntimeError("capture a training step through graphed_train_step (it allocates the NaN counter)")
                self._nan_subs += bad                 # counted on the device; check_device_status() reports it
                loss = torch.where(bad, torch.full_like(loss, 0.1), loss)
            elif bool(torch.isnan(loss)):
                print("loss is nan")
                loss.data = torch.ones_like(loss) * 0.1
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        xyz_sparse, xyz_dense, _ = self._unpack(batch)
        upratio = int(xyz_dense.shape[1] / xyz_sparse.shape[1])
        was = self.training
        if batch_idx == 0:
            from .dist import broadcast_buffers
            broadcast_buffers(self)              # multi-rank: rank 0's BN running statistics, as DDP would have them
        self.eval()                              # (the packed eval plan is re-used across validation batches: interpflow._engine)
        predict_x, logpx = self(xyz_sparse, upratio=upratio)
        cd = self.chamfer_loss2(predict_x, xyz_dense)
        self.train(was)
        return {"vloss": logpx.detach().cpu(), "CD": cd}

    def validation_epoch_end(self, batch):
        log_dict = {"vloss": torch.tensor([x["vloss"] * 1e-5 for x in batch]).sum().item(),
                    "CD": torch.tensor([float(x["CD"]) for x in batch]).sum().item()}
        self.log("CD", log_dict["CD"])
        self.log("vloss", log_dict["vloss"])
        self.epoch += 1
        return log_dict

    # ---- what Lightning's loop does around training_step, made explicit (and multi-GPU aware)
    def _sync_actnorm_init(self, batch) -> None:
        """ActNorm's data-dependent first-batch init (normalize.py:45-54) would differ per rank under batch sharding and
        gradient averaging never reconciles parameters: run the init forward once, then broadcast rank 0's module."""
        if all(b.actnorm.is_inited for b in self.network.flow_blocks):
            return
        from .dist import multi_rank
        if not multi_rank():
            return                      # single process: the init happens inside the first real forward, as in the reference
        from .dist import broadcast_module
        with torch.no_grad():
            sparse, dense, _ = self._unpack(batch)
            # this extra train-mode forward exists for ActNorm only: the BatchNorm buffers (running statistics, batch
            # counters) are put back afterwards, so the first real step leaves them exactly where a single-process run does
            bufs = list(self.network.buffers())
            keep = [b.clone() for b in bufs]
            self(sparse, upratio=int(dense.shape[1] / sparse.shape[1]))
            for b, k in zip(bufs, keep):
                b.copy_(k)
        broadcast_module(self)

    def check_device_status(self, raise_on_nan: bool = False) -> int:
        """What the device recorded since the last call, read at a point where the host synchronises anyway (end of an eager
        step, every few replays of a captured one, end of an epoch): a timed-out EMD grid barrier raises PuflowHipError; the
        number of NaN losses a CAPTURED step replaced by the reference's constant 0.1 (train_pu1k.py:71-73) is returned (and
        printed, like the reference's `loss is nan`)."""
        from .loss import check_emd_status
        from .train_ops import check_persist_status
        check_emd_status(next(self.parameters()).device)
        check_persist_status(next(self.parameters()).device)
        opt = getattr(self, "_fused_opt", None)
        if opt is not None:
            k = opt.skipped_updates()
            if k:
                print(f"optimizer: {k} update(s) skipped on the device (non-finite gradient norm; parameters and moments untouched)")
                if raise_on_nan:
                    raise RuntimeError(f"{k} training step(s) produced a non-finite gradient")
        n = 0
        if self._nan_subs is not None:
            n = int(self._nan_subs.item())
            if n:
                self._nan_subs.zero_()
                print(f"loss is nan ({n} captured step(s): replaced by 0.1, zero gradient)")
                if raise_on_nan:
                    raise RuntimeError(f"{n} training step(s) produced a NaN loss")
        return n

    def graphed_train_step(self, batch, optimizer: torch.optim.Optimizer, clip: float = 1e-2, warmup: int = 2):
        """`train_step` for a fixed batch shape captured in hipGraphs (puflow_amd/train_graph.py): returns
        `step(batch) -> loss` that replays forward + backward (+ clip + Adam) with one launch instead of ~3 500."""
        from .optim import FusedClipAdam
        from .train_graph import GraphedTrainStep
        if isinstance(optimizer, FusedClipAdam):
            self._fused_opt = optimizer                         # check_device_status reports its skipped updates
        return GraphedTrainStep(self, optimizer, batch, clip, warmup)

    def backward_seed(self, loss: Tensor) -> Tensor:
        """d loss / d loss = 1 as a tensor that exists already: `loss.backward()` would fill a new one each step (a launch)."""
        one = getattr(self, "_seed_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            if loss.is_cuda and torch.cuda.is_current_stream_capturing():
                return torch.ones_like(loss)
            one = torch.ones_like(loss)
            self._seed_one = one
        return one

    def train_step(self, batch, optimizer: torch.optim.Optimizer, clip: float = 1e-2) -> Tensor:
        self.train()
        self._sync_actnorm_init(batch)
        if self._bucket is None or self._bucket.flat.device != next(self.parameters()).device:
            self._bucket = FlatGradBucket(self.parameters())
        self._bucket.drop_grads()                               # = optimizer.zero_grad(set_to_none=True): no fill, no accumulate
        loss = self.training_step(batch, 0)
        loss.backward(self.backward_seed(loss))
        from .optim import FusedClipAdam
        if isinstance(optimizer, FusedClipAdam):
            self._fused_opt = optimizer
            self._bucket.all_reduce_mean(always_pack=True)      # ONE concatenation (+ ONE 3.2 MB RCCL all-reduce when multi-rank)
            optimizer.param_groups[0]["max_norm"] = clip
            optimizer.step_flat(self._bucket.flat)              # clip by the global norm + Adam: two launches
        else:
            self._bucket.all_reduce_mean()                      # multi-rank: ONE concatenation + ONE 3.2 MB RCCL all-reduce
            torch.nn.utils.clip_grad_norm_(self._bucket.params, clip, foreach=True)
            optimizer.step()
        if not torch.cuda.is_current_stream_capturing() and loss.is_cuda:
            self.check_device_status()                          # eager step: one word read (the NaN guard above read the loss already)
        return loss.detach()
