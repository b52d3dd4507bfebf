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

from ..losses._ops import get_ops, library_fallback


def _bf16_of(p):
    """bf16 image of a parameter: FlatParams.refresh_bf16() casts every parameter with ONE kernel per step; fall back
    to a per-call cast when the module is used outside the trainer"""
    if p is None:
        return None
    p16 = getattr(p, "_basd_bf16", None)
    return p16 if p16 is not None else p.to(torch.bfloat16)


def _bf16_compute(x) -> bool:
    """bf16 activations, or bf16 autocast on the device of ``x``"""
    if x.dtype == torch.bfloat16:
        return True
    dev = x.device.type
    return torch.is_autocast_enabled(dev) and torch.get_autocast_dtype(dev) == torch.bfloat16


def _bf16_transposed(weight, w16):
    """bf16 W^T for the input-gradient GEMM: the image FlatParams keeps (one launch per step for all layers), else a
    per-call transpose (<= 0.6 M elements)"""
    wt = getattr(weight, "_basd_bf16_t", None)
    return wt if wt is not None else w16.t().contiguous()


def _weight_grads(weight, bias, dy2, x2):
    """dW = dY^T X, db = sum dY.  With gradient sinks (views of the flat fp32 gradient buffer) the kernel accumulates
    straight into them -- no temporary, no AccumulateGrad add kernel -- and the data-parallel reducer is told that the
    slots are ready; (None, None) is returned to autograd then."""
    sink_w = getattr(weight, "_basd_grad", None)
    sink_b = getattr(bias, "_basd_grad", None) if bias is not None else None
    if sink_w is not None and (bias is None or sink_b is not None):
        get_ops().wgrad_bf16(dy2, x2, need_bias=bias is not None, out_w=sink_w, out_b=sink_b)
        for p in (weight, bias):
            ready = getattr(p, "_basd_ready", None) if p is not None else None
            if ready is not None:
                ready()
        return None, None
    return get_ops().wgrad_bf16(dy2, x2, need_bias=bias is not None)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x16 = x.to(torch.bfloat16)
        w16, b16 = _bf16_of(weight), _bf16_of(bias)
        ctx.save_for_backward(x16, w16)
        ctx.weight, ctx.bias = weight, bias
        ctx.x_dtype = x.dtype
        ops = get_ops()
        if ops.gemm_supported(w16.shape[0], w16.shape[1]):
            return ops.gemm_bf16(x16, w16, b16)             # hand-written MFMA GEMM, bias in the epilogue
        library_fallback("linear forward", f"N={w16.shape[0]} K={w16.shape[1]}")
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
                gx = ops.gemm_bf16(g16, _bf16_transposed(weight, w16)).to(ctx.x_dtype)
            else:
                library_fallback("linear input gradient", f"N={w16.shape[1]} K={w16.shape[0]}")
                gx = (g16 @ w16).to(ctx.x_dtype)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            gw, gb = _weight_grads(weight, bias, g16.reshape(-1, g16.shape[-1]), x16.reshape(-1, x16.shape[-1]))
        return gx, gw, gb


class _MlpFn(torch.autograd.Function):
    """fc1 -> GELU -> fc2 of a trained block (timm Mlp): the GELU rides in the epilogue of the fc1 GEMM (which also
    stores the pre-activation) and its backward in the epilogue of fc2's input-gradient GEMM: two elementwise passes
    over the [B T, hidden] activations less per block and direction."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ops = get_ops()
        x16 = x.to(torch.bfloat16)
        w1_16, w2_16 = _bf16_of(w1), _bf16_of(w2)
        pre, act = ops.gemm_gelu_fwd(x16, w1_16, _bf16_of(b1))
        y = ops.gemm_bf16(act, w2_16, _bf16_of(b2))
        ctx.save_for_backward(x16, pre, act, w1_16, w2_16)
        ctx.params = (w1, b1, w2, b2)
        ctx.x_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        x16, pre, act, w1_16, w2_16 = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        ops = get_ops()
        g16 = g.to(torch.bfloat16).contiguous()
        hidden, d_in = w1_16.shape
        dpre = ops.gemm_gelu_bwd(g16, _bf16_transposed(w2, w2_16), pre)     # (dY W2) * gelu'(pre)
        gw2, gb2 = _weight_grads(w2, b2, g16.reshape(-1, g16.shape[-1]), act.reshape(-1, hidden))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.gemm_bf16(dpre, _bf16_transposed(w1, w1_16)).to(ctx.x_dtype)
        gw1, gb1 = _weight_grads(w1, b1, dpre.reshape(-1, hidden), x16.reshape(-1, d_in))
        return gx, gw1, gb1, gw2, gb2


def fused_mlp_ok(x, fc1: "BasdLinear", fc2: "BasdLinear") -> bool:
    """both layers trained, on device activations, every GEMM of forward and backward tiled by the own kernels"""
    ops = get_ops()
    return (ops.handles(x) and torch.is_grad_enabled() and x.dim() >= 2 and x.numel() // x.shape[-1] >= 64
            and fc1.weight.requires_grad and fc2.weight.requires_grad
            and fc1.bias is not None and fc2.bias is not None and fc1.bias.requires_grad and fc2.bias.requires_grad
            and ops.gemm_supported(fc1.out_features, fc1.in_features)        # fc1 forward
            and ops.gemm_supported(fc2.out_features, fc2.in_features)        # fc2 forward
            and ops.gemm_supported(fc2.in_features, fc2.out_features)        # fc2 input gradient (+ GELU backward)
            and ops.gemm_supported(fc1.in_features, fc1.out_features)        # fc1 input gradient
            and ops.wgrad_supported(fc1.out_features, fc1.in_features)
            and ops.wgrad_supported(fc2.out_features, fc2.in_features))


def fused_mlp(x, fc1: "BasdLinear", fc2: "BasdLinear"):
    with torch.autocast(device_type=x.device.type, enabled=False):
        return _MlpFn.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)


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
        if (get_ops().handles(x) and not torch.is_grad_enabled() and x.dim() >= 2
                and get_ops().gemm_supported(self.out_features, self.in_features) and _bf16_compute(x)):
            # evaluation of a TRAINED layer (fp32 master weights under bf16 autocast): same kernel, weights cast per
            # call (the optimizer's eval / train switch moves the weights, the step's bf16 images may be stale here)
            return get_ops().gemm_bf16(x.to(torch.bfloat16), self.weight.to(torch.bfloat16),
                                       None if self.bias is None else self.bias.to(torch.bfloat16))
        if get_ops().handles(x) and not getattr(self, "library_ok", False):
            library_fallback("linear", f"{self.in_features}->{self.out_features} dtype={x.dtype} rows="
                                       f"{x.numel() // max(1, x.shape[-1])} grad={torch.is_grad_enabled()}")
        return F.linear(x, self.weight, self.bias)

    def fused_inference_ok(self, x) -> bool:
        """frozen bf16 layer on device activations (the teacher): hand-written GEMM with the bias (and, from Mlp, the
        GELU) in its epilogue"""
        return (not torch.is_grad_enabled() and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16
                and (self.bias is None or self.bias.dtype == torch.bfloat16) and x.dim() >= 2
                and get_ops().handles(x) and get_ops().gemm_supported(self.out_features, self.in_features))
