"""Training entry point: what `train(phase, checkpoint_path, begin_checkpoint)` + `pl.Trainer.fit` do in the reference's
`modules/discrete/train_pu1k.py:124-184` (and its pugan / pugeo twins), without Lightning:

  * warm start from `begin_checkpoint` + `set_to_initialized_state()`                     (train_pu1k.py:166-168)
  * epochs of `TrainerModule.train_step` (forward -> loss -> backward -> one RCCL all-reduce -> clip 1e-2 -> Adam)
  * a validation pass per epoch (`validation_step` / `validation_epoch_end`), whose summed CD drives
    `ReduceLROnPlateau` - the reference returns the scheduler under the key 'scheduler' (train_pu1k.py:51), which Lightning
    ignores, so there it is never stepped; here it is
  * the end-of-run `state_dict` save, guarded like the reference: only for complete runs of more than 10 epochs, to
    `<checkpoint_path minus .ckpt>-epoch<N>.ckpt`                                          (train_pu1k.py:172-176)

One process per GPU under `torchrun` (batches are sharded by the data object, gradients averaged by one all-reduce);
rank 0 saves.  CLI:  python -m puflow_amd.train --dataset pu1k --data data/pu1k_...h5 [--synthetic] [--max_epochs 100]
"""
from __future__ import annotations

import argparse
import os
from typing import Iterable, Optional

import torch

from .trainer import TrainerModule, default_cfg

LOSS_MIX = {"pu1k": "pu1k", "pugeo": "pugeo", "pugan": "pugan"}
DEFAULT_CKPT = {"pu1k": "runs/ckpt/puflow-pu1k.ckpt", "pugeo": "runs/ckpt/puflow-pugeo.ckpt", "pugan": "runs/ckpt/puflow-pugan.ckpt"}


def model_specific_args() -> argparse.ArgumentParser:
    """The reference's flags and defaults (train_pu1k.py:109-121) plus what its script hard-codes."""
    p = argparse.ArgumentParser()
    p.add_argument("--net", type=str, default="UpsamplingFlow")
    p.add_argument("--learning_rate", default=1e-3, type=float)
    p.add_argument("--sched_patience", default=10, type=int)
    p.add_argument("--sched_factor", default=0.5, type=float)
    p.add_argument("--seed", default=2021, type=int)
    p.add_argument("--dataset", default="pu1k", choices=sorted(LOSS_MIX))
    p.add_argument("--data", default=None, help=".h5 (needs h5py) or .npz with poisson_<n> arrays")
    p.add_argument("--synthetic", action="store_true", help="in-memory surface patches instead of a file")
    p.add_argument("--batch_size", default=32, type=int)
    p.add_argument("--max_epochs", default=100, type=int)
    p.add_argument("--val_batches", default=400, type=int)
    p.add_argument("--checkpoint_path", default=None)
    p.add_argument("--begin_checkpoint", default=None)
    p.add_argument("--sync_batchnorm", action="store_true")
    p.add_argument("--graph", action="store_true", help="replay the training step from a hipGraph per batch shape")
    return p


def _dist():
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_rank(), d.get_world_size()
    return 0, 1


def _batch_key(batch):
    vals = batch.values() if isinstance(batch, dict) else batch
    return tuple(tuple(v.shape) for v in vals if isinstance(v, torch.Tensor))


def fit(module: TrainerModule, train_data: Iterable, val_data: Optional[Iterable], max_epochs: int,
        clip: float = 1e-2, log=print, graph: bool = False) -> dict:
    """The epoch loop of `pl.Trainer.fit` for this module.  Returns {'epochs', 'lr' (per epoch), 'CD' (per epoch)}.
    graph=True (GPU, fused optimizer): every batch shape is captured once as a hipGraph (`TrainerModule.graphed_train_step`;
    the capture's single warm-up step is that batch's optimisation step) and replayed for the batches that follow."""
    opt_cfg = module.configure_optimizers()
    optimizer, scheduler = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    hist = {"epochs": 0, "lr": [], "CD": [], "loss": []}
    rank, _ = _dist()
    graphs = {}
    use_graph = graph and next(module.parameters()).is_cuda
    for epoch in range(max_epochs):
        module.train()
        last = None
        for bi, batch in enumerate(train_data):
            if use_graph:
                key = _batch_key(batch)
                if key not in graphs:
                    graphs[key] = module.graphed_train_step(batch, optimizer, clip, warmup=1)
                    last = graphs[key].warmup_loss
                else:
                    last = graphs[key](batch)
                continue
            last = module.train_step(batch, optimizer, clip)
        hist["loss"].append(float(last) if last is not None else float("nan"))
        if next(module.parameters()).is_cuda:
            module.check_device_status()          # end of epoch, the host just synchronised: EMD barrier time-outs raise, NaN substitutions are printed
        if val_data is not None:
            outs = [module.validation_step(b, i) for i, b in enumerate(val_data)]
            cd = module.validation_epoch_end(outs)["CD"]
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                t = torch.tensor([cd], dtype=torch.float64, device=next(module.parameters()).device)
                torch.distributed.all_reduce(t)                     # every rank steps the scheduler on the same number
                cd = float(t.item())
            scheduler.step(cd)                                      # monitor: 'CD' (configure_optimizers)
            hist["CD"].append(cd)
        hist["lr"].append(optimizer.param_groups[0]["lr"])
        hist["epochs"] = epoch + 1
        if rank == 0 and log is not None:
            log(f"epoch {epoch}: loss {hist['loss'][-1]:.6f}" + (f"  CD {hist['CD'][-1]:.6f}" if hist["CD"] else "") +
                f"  lr {hist['lr'][-1]:.2e}")
    return hist


def train(phase: str = "Train", checkpoint_path: Optional[str] = None, begin_checkpoint: Optional[str] = None, cfg=None,
          train_data: Optional[Iterable] = None, val_data: Optional[Iterable] = None, max_epochs: int = 100,
          dataset: str = "pu1k", device: Optional[str] = None, log=print, module: Optional[TrainerModule] = None,
          graph: bool = False):
    """Reference signature `train(phase, checkpoint_path, begin_checkpoint)` (train_pu1k.py:124); the keyword arguments
    replace what the reference hard-codes (datamodule, trainer_config).  Returns (module, history)."""
    cfg = cfg or default_cfg()
    torch.manual_seed(int(getattr(cfg, "seed", 2021)))
    dev = device or (f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}" if torch.cuda.is_available() else "cpu")
    module = (module or TrainerModule(cfg, loss_mix=LOSS_MIX[dataset])).to(dev)
    hist = None
    if phase == "Train":
        if begin_checkpoint is not None:
            module.network.load_state_dict(torch.load(begin_checkpoint, map_location=dev))
            module.network.set_to_initialized_state()
        if train_data is None:
            raise ValueError("train(): no training data (pass train_data or use the CLI's --data / --synthetic)")
        hist = fit(module, train_data, val_data, max_epochs, log=log, graph=graph)
        rank, _ = _dist()
        if checkpoint_path is not None and hist["epochs"] == max_epochs and max_epochs > 10 and rank == 0:
            save_path = checkpoint_path.replace(".ckpt", f"-epoch{max_epochs}.ckpt")
            os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
            torch.save(module.network.state_dict(), save_path)
            if log is not None:
                log(f"Model has been saved to {save_path}")
    return module, hist


def main(argv=None) -> None:
    from .data import SyntheticPatchData, patch_data_from_file
    a = model_specific_args().parse_args(argv)
    from ._host import limit_host_threads
    limit_host_threads()                       # torch's OpenMP pool within the container's CPU quota (puflow_amd/_host.py)
    if "RANK" in os.environ and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    rank, world = _dist()
    dev = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}" if torch.cuda.is_available() else "cpu"
    if dev.startswith("cuda"):
        torch.cuda.set_device(dev)
    cfg = default_cfg(learning_rate=a.learning_rate, sched_patience=a.sched_patience, sched_factor=a.sched_factor,
                      seed=a.seed, sync_batchnorm=a.sync_batchnorm)
    kw = dict(batch_size=a.batch_size, num_point_patch=256, device=dev, seed=a.seed, rank=rank, world=world,
              is_augment=True, jitter_sigma=0.01, jitter_max=0.03)            # train_pu1k.py:132-141
    if a.synthetic or a.data is None:
        tr = SyntheticPatchData(num_patches=8 * a.batch_size, up_ratio=4, **kw)
        va = SyntheticPatchData(num_patches=2 * a.batch_size, up_ratio=4, **{**kw, "is_augment": False, "seed": a.seed + 7})
    else:
        tr = patch_data_from_file(a.data, up_ratio=4, **kw)
        va = patch_data_from_file(a.data, up_ratio=4, num_batches=a.val_batches, **{**kw, "is_augment": False})
    train("Train", a.checkpoint_path or DEFAULT_CKPT[a.dataset], a.begin_checkpoint, cfg, tr, va, a.max_epochs, a.dataset, dev,
          graph=a.graph)


if __name__ == "__main__":
    main()
