"""nn.Linear whose weight/bias gradient runs on the hand-written split-M MFMA kernel.

Forward and input-gradient GEMMs stay on hipBLASLt this round (well served: 1 PF/s on the
teacher shapes); the weight gradient ``dW = dY^T X`` reduces over B*T ~ 50k rows into a
<= 768 x 768 output, which the library runs as a 12-workgroup launch.
The function casts the fp32 master weight to bf16 itself and returns fp32 ``dW`` / ``db``
directly, so autograd adds them into the flat fp32 gradient buffer without a cast kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..losses._ops import get_ops, is_emulated


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x16 = x.to(torch.bfloat16)
        w16 = weight.to(torch.bfloat16)
        ctx.save_for_backward(x16, w16)
        ctx.has_bias = bias is not None
        ctx.x_dtype = x.dtype
        return F.linear(x16, w16, None if bias is None else bias.to(torch.bfloat16))

    @staticmethod
    def backward(ctx, g):
        x16, w16 = ctx.saved_tensors
        g16 = g.to(torch.bfloat16).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (g16 @ w16).to(ctx.x_dtype)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = get_ops().wgrad_bf16(g16.reshape(-1, g16.shape[-1]), x16.reshape(-1, x16.shape[-1]),
                                          need_bias=ctx.has_bias)
        return gx, gw, gb


class BasdLinear(nn.Linear):
    """Drop-in nn.Linear (same parameters / state_dict keys)."""

    def forward(self, x):
        # device tensors only (a CPU module is plain nn.Linear: used by the CPU baseline and by the
        # plumbing tests, which install the test-only kernel emulation)
        if ((x.is_cuda or is_emulated()) and self.weight.requires_grad and torch.is_grad_enabled() and x.dim() >= 2
                and get_ops().wgrad_supported(self.out_features, self.in_features)
                and x.numel() // x.shape[-1] >= 64):
            with torch.autocast(device_type=x.device.type, enabled=False):
                return _LinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
