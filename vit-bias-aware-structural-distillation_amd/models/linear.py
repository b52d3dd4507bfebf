"""nn.Linear on the hand-written MFMA kernels: forward and input gradient on ``basd_gemm_bf16``
(256 x 256 tiles, LDS-DMA operand ring, bias / GELU epilogue), weight / bias gradient on the
split-M ``basd_wgrad_bf16``.  Shapes the kernels do not tile (the 1000-class head) stay on the
library.  The weight gradient ``dW = dY^T X`` reduces over B*T ~ 50k rows into a
<= 768 x 768 output, which the library runs as a 12-workgroup launch.
The function casts the fp32 master weight to bf16 itself and returns fp32 ``dW`` / ``db``
directly, so autograd adds them into the flat fp32 gradient buffer without a cast kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..losses._ops import get_ops


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x16 = x.to(torch.bfloat16)
        # FlatParams.refresh_bf16() casts every parameter with ONE kernel per step; fall back to a
        # per-call cast when the module is used outside the trainer
        w16 = getattr(weight, "_basd_bf16", None)
        if w16 is None:
            w16 = weight.to(torch.bfloat16)
        b16 = None
        if bias is not None:
            b16 = getattr(bias, "_basd_bf16", None)
            if b16 is None:
                b16 = bias.to(torch.bfloat16)
        ctx.save_for_backward(x16, w16)
        ctx.weight, ctx.bias = weight, bias
        ctx.x_dtype = x.dtype
        ops = get_ops()
        if ops.gemm_supported(w16.shape[0], w16.shape[1]):
            return ops.gemm_bf16(x16, w16, b16)             # hand-written MFMA GEMM, bias in the epilogue
        return F.linear(x16, w16, b16)

    @staticmethod
    def backward(ctx, g):
        x16, w16 = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        g16 = g.to(torch.bfloat16).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            ops = get_ops()
            if ops.gemm_supported(w16.shape[1], w16.shape[0]):
                # dX = dY W as the same "NT" kernel on the transposed weight (<= 0.6 M elements: the copy is noise)
                gx = ops.gemm_bf16(g16, w16.t().contiguous()).to(ctx.x_dtype)
            else:
                gx = (g16 @ w16).to(ctx.x_dtype)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            sink_w = getattr(weight, "_basd_grad", None)
            sink_b = getattr(bias, "_basd_grad", None) if bias is not None else None
            dy2, x2 = g16.reshape(-1, g16.shape[-1]), x16.reshape(-1, x16.shape[-1])
            if sink_w is not None and (bias is None or sink_b is not None):
                # accumulate straight into the flat fp32 gradient buffer: no temporary, no
                # AccumulateGrad add kernel; tell the data-parallel reducer the slots are ready
                get_ops().wgrad_bf16(dy2, x2, need_bias=bias is not None, out_w=sink_w, out_b=sink_b)
                for p in (weight, bias):
                    ready = getattr(p, "_basd_ready", None) if p is not None else None
                    if ready is not None:
                        ready()
            else:
                gw, gb = get_ops().wgrad_bf16(dy2, x2, need_bias=bias is not None)
        return gx, gw, gb


class BasdLinear(nn.Linear):
    """Drop-in nn.Linear (same parameters / state_dict keys)."""

    def forward(self, x):
        # device tensors only (a CPU module is plain nn.Linear: used by the CPU baseline and by the
        # plumbing tests, which install the test-only kernel emulation)
        if (get_ops().handles(x) and self.weight.requires_grad and torch.is_grad_enabled() and x.dim() >= 2
                and get_ops().wgrad_supported(self.out_features, self.in_features)
                and x.numel() // x.shape[-1] >= 64):
            with torch.autocast(device_type=x.device.type, enabled=False):
                return _LinearFn.apply(x, self.weight, self.bias)
        if self.fused_inference_ok(x):
            return get_ops().gemm_bf16(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)

    def fused_inference_ok(self, x) -> bool:
        """frozen bf16 layer on device activations (the teacher): hand-written GEMM with the bias (and, from Mlp, the
        GELU) in its epilogue"""
        return (not torch.is_grad_enabled() and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16
                and (self.bias is None or self.bias.dtype == torch.bfloat16) and x.dim() >= 2
                and get_ops().handles(x) and get_ops().gemm_supported(self.out_features, self.in_features))
