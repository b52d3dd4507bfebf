"""Schedule-Free AdamW on flat fp32 buffers, one fused HIP kernel per step.

Restates ``schedulefree==1.4.1`` ``AdamWScheduleFree`` (the reference's optimizer,
``src/training/trainer.py:54-58``; defaults betas (0.9, 0.999), eps 1e-8,
warmup 0, r 0, weight_lr_power 2).  schedulefree is not installed in the build
image, so parity with it is UNPINNED; ``tests/test_optim.py`` pins this
restatement to a hand-written per-element trace of the published update rule.

All parameters live in ONE flat fp32 buffer (``FlatParams``): parameters and
gradients are views into it, which also gives the data-parallel reducer its
contiguous buckets and makes ``zero_grad`` one memset.
"""
from __future__ import annotations

import torch

from ..losses._ops import get_ops


class FlatParams:
    """Re-homes ``params`` (and their grads) as views of two flat fp32 buffers."""

    def __init__(self, params, align: int = 64):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        self.offsets, self.numel = offs, total
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.data16 = None
        self.data16_t, self._t_table = None, []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1).float())
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)

    def enable_bf16_shadow(self, params=None, transposed=None) -> None:
        """Keep a bf16 copy of the whole buffer (refreshed by ONE cast kernel per step) and expose, on
        each listed parameter, ``_basd_bf16`` (its bf16 view) and ``_basd_grad`` (its fp32 gradient
        slot) for kernels that read the half-precision weight / accumulate gradients directly.
        ``transposed``: 2-D weights that also get ``_basd_bf16_t`` (bf16 W^T, the operand of the input-gradient
        GEMM), all refreshed by one more launch."""
        self.data16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.data.device)
        chosen = None if params is None else {id(p) for p in params}
        want_t = set() if transposed is None else {id(p) for p in transposed}
        t_total = 0
        for p, o in zip(self.params, self.offsets):
            if chosen is None or id(p) in chosen:
                n = p.numel()
                p._basd_bf16 = self.data16[o:o + n].view(p.shape)
                p._basd_grad = self.grad[o:o + n].view(p.shape)
            if id(p) in want_t and p.dim() == 2:
                self._t_table.append((o, t_total, p.shape[0], p.shape[1]))
                t_total += (p.numel() + 63) // 64 * 64
        if self._t_table:
            self.data16_t = torch.empty(t_total, dtype=torch.bfloat16, device=self.data.device)
            by_offset = {o: p for p, o in zip(self.params, self.offsets)}
            for src, dst, rows, cols in self._t_table:
                by_offset[src]._basd_bf16_t = self.data16_t[dst:dst + rows * cols].view(cols, rows)
        self.refresh_bf16()

    def refresh_bf16(self) -> None:
        if self.data16 is not None:
            self.data16.copy_(self.data)
        if self.data16_t is not None:
            get_ops().transpose_table(self.data, self.data16_t, self._t_table)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach if autograd replaced a view
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class AdamWScheduleFree:
    def __init__(self, flat: FlatParams, lr: float = 0.0025, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, warmup_steps: int = 0, r: float = 0.0, weight_lr_power: float = 2.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.warmup_steps, self.r, self.weight_lr_power = warmup_steps, r, weight_lr_power
        self.z = flat.data.clone()
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.k = 0
        self.weight_sum = 0.0
        self.lr_max = -1.0
        self.train_mode = False
        self.scheduled_lr = lr

    # -- mode switches (reference trainer.py:180,184; train.py:153) ----------
    def train(self):
        if not self.train_mode:
            get_ops().lerp_(self.flat.data, self.z, 1.0 - self.betas[0])      # x -> y
            self.train_mode = True

    def eval(self):
        if self.train_mode:
            get_ops().lerp_(self.flat.data, self.z, 1.0 - 1.0 / self.betas[0])   # y -> x
            self.train_mode = False

    def step_scalars(self):
        k = self.k
        sched = (k + 1) / self.warmup_steps if k < self.warmup_steps else 1.0
        bias_correction2 = 1.0 - self.betas[1] ** (k + 1)
        lr = self.lr * sched
        self.scheduled_lr = lr
        self.lr_max = max(lr, self.lr_max)
        weight = ((k + 1) ** self.r) * (self.lr_max ** self.weight_lr_power)
        self.weight_sum += weight
        ckp1 = weight / self.weight_sum if self.weight_sum != 0 else 0.0
        return lr, ckp1, bias_correction2

    def step(self):
        if not self.train_mode:
            raise RuntimeError("AdamWScheduleFree.step() called in eval mode; call optimizer.train() first")
        lr, ckp1, bc2 = self.step_scalars()
        get_ops().sf_adamw_step(self.flat.data, self.flat.grad, self.z, self.exp_avg_sq, lr=lr,
                                beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                                weight_decay=self.weight_decay, ckp1=ckp1, bias_correction2=bc2)
        self.k += 1

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        return {"z": self.z, "exp_avg_sq": self.exp_avg_sq, "k": self.k, "weight_sum": self.weight_sum,
                "lr_max": self.lr_max, "train_mode": self.train_mode, "params": self.flat.data}

    def load_state_dict(self, sd):
        self.z.copy_(sd["z"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.flat.data.copy_(sd["params"])
        self.k, self.weight_sum, self.lr_max = sd["k"], sd["weight_sum"], sd["lr_max"]
        self.train_mode = sd["train_mode"]
