"""AdamW + StepLR of the reference's train loop (multi_train_MDViT.py:88-95: optim.AdamW(lr, weight_decay),
lr_scheduler.StepLR(step_size=50, gamma=0.5)) as ONE HIP launch per step over every parameter.

FusedAdamW works on a parallel.GradAccumulator: the gradients are the accumulator's flat bucket views, the two moments
are flat buffers with the same layout, and a device table of pointers drives the kernel -- one table row per 32768-element CHUNK
of a parameter (the library slices tables of more than 65535 rows over several launches); the step counter and the
learning rate live in device memory (HIP-graph replay safe).

Deviation from torch.optim.AdamW, by construction: the kernel updates EVERY parameter of the accumulator on every step (gradient =
the bucket view, zero if nothing was accumulated), so a parameter that received no gradient at all in a step is still weight-decayed
and its moments still decay -- torch skips parameters whose .grad is None.  In the train steps of this package every parameter is
reached by the summed loss on every step (checked by the golden gradient-norm tests: no zero-norm entries), so the two agree there."""
from __future__ import annotations

import ctypes as C

import torch

from . import ops
from ._lib import call
from .parallel import GradAccumulator


class FusedAdamW:
    _CHUNK = 32768

    def __init__(self, accumulator: GradAccumulator, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, zero_grad: bool = False):
        self.acc = accumulator
        self.params = accumulator.params
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FusedAdamW needs CUDA parameters (there is no CPU fallback)")
        dev = self.params[0].device
        self.betas, self.eps, self.weight_decay, self.zero_grad_after = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay), bool(zero_grad)
        self.exp_avg = [torch.zeros_like(b) for b in accumulator.reducer.buckets]
        self.exp_avg_sq = [torch.zeros_like(b) for b in accumulator.reducer.buckets]
        # one table row per CHUNK of a parameter (<= _CHUNK elements, a multiple of 4 so that a 16-byte aligned tensor stays on the float4 path): the kernel gives every row
        # the same number of workgroups, and with one row per TENSOR the 64 workgroups of the largest weight (4.7 M elements of ~31 M) were still walking it when everything
        # else had long finished -- 236 us per step at 4.2 TB/s, alone on the GPU between the last gradient and the next forward
        rows = []
        for p, g in zip(self.params, accumulator.views):
            bi = accumulator.reducer._bucket_of[p]
            off = g.data_ptr() - accumulator.reducer.buckets[bi].data_ptr()
            if not p.is_contiguous():
                raise RuntimeError("FusedAdamW needs contiguous parameters")
            m0, v0, n = self.exp_avg[bi].data_ptr() + off, self.exp_avg_sq[bi].data_ptr() + off, p.numel()
            for c0 in range(0, n, self._CHUNK):
                rows.append([p.data_ptr() + 4 * c0, g.data_ptr() + 4 * c0, m0 + 4 * c0, v0 + 4 * c0, min(self._CHUNK, n - c0)])
        self.n_rows = len(rows)
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.lr_dev = torch.tensor([float(lr)], dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": float(lr), "params": self.params}]      # what lr schedulers read and write
        self._lr_host = float(lr)
        biggest = min(self._CHUNK, max(p.numel() for p in self.params))
        self.blocks = int(min(8, max(1, (biggest // 4 + 255) // 256)))

    def set_lr(self, lr: float):
        """host-side schedule -> device scalar (one tiny H2D copy, outside any captured region)"""
        self._lr_host = float(lr)
        self.param_groups[0]["lr"] = float(lr)
        self.lr_dev.fill_(float(lr))

    def zero_grad(self, set_to_none: bool = True):
        self.acc.zero()

    def step(self):
        if self.param_groups[0]["lr"] != self._lr_host and not torch.cuda.is_current_stream_capturing():
            self.set_lr(self.param_groups[0]["lr"])          # a torch scheduler wrote the new rate into param_groups
        call("mdvit_adamw_step", C.c_void_p(self.table.data_ptr()), self.n_rows, self.blocks, C.c_void_p(self.lr_dev.data_ptr()),
             C.c_void_p(self.step_dev.data_ptr()), self.betas[0], self.betas[1], self.eps, self.weight_decay, int(self.zero_grad_after),
             ops._stream())
        # the kernel wrote the parameters through raw pointers: Tensor._version did not move, so the cached W^T copies and weight
        # planes of the GEMMs are invalidated HERE (they are rebuilt by ops.refresh_transposes() at the start of the next step,
        # or lazily by the first GEMM that touches a weight)
        ops.mark_weights_updated()


class StepLR:
    """lr = base_lr * gamma ** (epoch // step_size)   (optim.lr_scheduler.StepLR(optimizer, 50, 0.5), multi_train_MDViT.py:95)"""

    def __init__(self, optimizer, step_size: int = 50, gamma: float = 0.5):
        self.opt, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        lr = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)
        if hasattr(self.opt, "set_lr"):
            self.opt.set_lr(lr)
        else:
            for g in self.opt.param_groups:
                g["lr"] = lr

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]
