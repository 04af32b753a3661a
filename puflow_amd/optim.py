"""FusedClipAdam: gradient clipping by the global L2 norm + Adam as two HIP launches for the whole model (csrc/optim.hip).

Same arithmetic as `torch.nn.utils.clip_grad_norm_(params, max_norm)` followed by `torch.optim.Adam(lr, betas, eps).step()`
(the reference: Lightning `gradient_clip_val=1e-2` + `torch.optim.Adam`, train_pu1k.py:46,149).  It is a
`torch.optim.Optimizer` (one parameter group, `param_groups[0]["lr"]` is what `ReduceLROnPlateau` changes), so the rest of the
training entry does not notice.  Why: inside a captured training step PyTorch's capturable Adam runs ~550 per-tensor kernels
(2.4 ms); this is 1 concatenation + 2 launches.

The gradients are consumed as ONE flat buffer in parameter order (`FlatGradBucket.pack()` / `.flat`); the moments are flat
buffers of the same layout; the parameters stay where they are.
"""
from __future__ import annotations

import torch

from . import _lib

CHUNK = 4096


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, max_norm=max_norm))
        if len(self.param_groups) != 1:
            raise ValueError("FusedClipAdam: one parameter group (the reference uses one)")
        self.params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise _lib.PuflowHipError("FusedClipAdam runs on the GPU only: move the module before building the optimizer")
        rows, off = [], 0
        for t, p in enumerate(self.params):
            n = p.numel()
            for lo in range(0, n, CHUNK):
                rows.append((t, lo, min(CHUNK, n - lo), off + lo))
            off += n
        self.numel = off
        self._chunks = torch.tensor(rows, dtype=torch.int32).to(dev)
        self._ptrs = torch.tensor([p.data_ptr() for p in self.params], dtype=torch.int64).to(dev)
        self._ptr_sig = tuple(p.data_ptr() for p in self.params)
        f32 = dict(dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, **f32)
        self.exp_avg_sq = torch.zeros(off, **f32)
        self.step_t = torch.zeros(1, **f32)
        self.lr_t = torch.full((1,), float(lr), **f32)
        self._lr_host = float(lr)
        # [clip coefficient, gradient norm, this update skipped (non-finite norm), number of skipped updates] - csrc/optim.hip
        self.coef = torch.zeros(4, **f32)
        self._partial = torch.empty(len(rows), dtype=torch.float64, device=dev)
        self._counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lib = _lib.load()

    def sync_lr(self) -> None:
        """param_groups[0]['lr'] (what a scheduler edits) -> the device scalar the kernels read (one fill when it changed)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_t.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def step_flat(self, flat_grad: torch.Tensor) -> None:
        """One clipped Adam update from the flat gradient buffer (modified in place: it holds the clipped gradient after)."""
        if flat_grad.numel() != self.numel or not flat_grad.is_contiguous():
            raise ValueError("FusedClipAdam.step_flat: flat gradient of the wrong size")
        if tuple(p.data_ptr() for p in self.params) != self._ptr_sig:          # parameters were moved (.to(), load): re-point
            self._ptrs.copy_(torch.tensor([p.data_ptr() for p in self.params], dtype=torch.int64))
            self._ptr_sig = tuple(p.data_ptr() for p in self.params)
        g = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        _lib.check(self.lib.pf_clip_adam(flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                         self._ptrs.data_ptr(), self._chunks.data_ptr(), self._chunks.shape[0],
                                         self.lr_t.data_ptr(), self.step_t.data_ptr(), float(g["betas"][0]), float(g["betas"][1]),
                                         float(g["eps"]), float(g["max_norm"]), self._partial.data_ptr(),
                                         self._counter.data_ptr(), self.coef.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "pf_clip_adam")
        # the kernel writes the parameters through raw pointers: tell torch (autograd's saved-tensor checks, and the packed
        # eval plan of PointInterpFlow, which is keyed on the version counters)
        torch._C._increment_version(self.params)

    @torch.no_grad()
    def step_table(self, grad_table: torch.Tensor) -> None:
        """One clipped Adam update with the gradients where autograd left them: grad_table = device int64 [n parameters], the
        address of every parameter's (contiguous fp32) gradient in `self.params` order - `grad_table_of()` builds it.  For a
        captured training step: the gradients' addresses are the same at every replay, so the table is written once and the
        step needs no concatenation launches (pf_clip_adam_ptrs)."""
        if grad_table.dtype != torch.int64 or grad_table.numel() != len(self.params) or not grad_table.is_cuda:
            raise ValueError("FusedClipAdam.step_table: one int64 device address per parameter")
        if tuple(p.data_ptr() for p in self.params) != self._ptr_sig:
            self._ptrs.copy_(torch.tensor([p.data_ptr() for p in self.params], dtype=torch.int64))
            self._ptr_sig = tuple(p.data_ptr() for p in self.params)
        g = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        _lib.check(self.lib.pf_clip_adam_ptrs(grad_table.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                              self._ptrs.data_ptr(), self._chunks.data_ptr(), self._chunks.shape[0],
                                              self.lr_t.data_ptr(), self.step_t.data_ptr(), float(g["betas"][0]), float(g["betas"][1]),
                                              float(g["eps"]), float(g["max_norm"]), self._partial.data_ptr(),
                                              self._counter.data_ptr(), self.coef.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "pf_clip_adam_ptrs")
        torch._C._increment_version(self.params)

    def grad_table_of(self, zeros: torch.Tensor):
        """Host list of gradient addresses for `step_table` (parameters without a gradient -> `zeros`, a zero fp32 buffer at least
        as long as the longest parameter), or None when a gradient is not a contiguous fp32 tensor of its parameter's size."""
        out = []
        for p in self.params:
            gr = p.grad
            if gr is None:
                if zeros.numel() < p.numel():
                    return None
                out.append(zeros.data_ptr())
            elif gr.dtype != torch.float32 or not gr.is_contiguous() or gr.numel() != p.numel() or gr.device != p.device:
                return None
            else:
                out.append(gr.data_ptr())
        return out

    def skipped_updates(self, reset: bool = True) -> int:
        """Updates the kernel SKIPPED since the last call because the global gradient norm was not finite (parameters, moments
        and step counter were left untouched).  One device read: call where the host synchronises anyway."""
        n = int(self.coef[3].item())
        if n and reset:
            self.coef[3].zero_()
        return n

    @torch.no_grad()
    def step(self, closure=None):
        """torch.optim interface: packs the parameters' .grad (missing ones count as zero) and updates."""
        loss = closure() if closure is not None else None
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
        self.step_flat(flat)
        return loss

    # the moments live outside `self.state`: carry them through checkpoints explicitly
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_t.clone()}
        return sd

    def load_state_dict(self, sd):
        fused = sd.get("fused")
        super().load_state_dict({k: v for k, v in sd.items() if k != "fused"})
        if fused is not None:
            self.exp_avg.copy_(fused["exp_avg"]); self.exp_avg_sq.copy_(fused["exp_avg_sq"]); self.step_t.copy_(fused["step"])
        self._lr_host = float("nan")
