"""Plain ViT / DeiT / DINOv2-style vision transformer for the BASD train step.

The reference takes its blocks from ``timm==1.0.24`` (``src/train.py:51``,
``src/models/teacher.py:118``), which is not part of the reference repository
and not installed here: this is an own implementation of the public ViT
definition, constrained by what the reference itself assumes of it --
``blocks`` container, ``attn.qkv`` single Linear with ``reshape(B,N,3,nh,hd)``
split order, ``attn.num_heads``, scale ``hd**-0.5``, CLS token at index 0
(teacher.py:33-37,46-50,156-157; trainer.py:29).  Parameter names follow timm
(``cls_token``, ``pos_embed``, ``patch_embed.proj``, ``blocks.i.norm1``,
``attn.qkv``, ``attn.proj``, ``ls1.gamma``, ``mlp.fc1`` ...) so that
``model_state_dict`` files are interchangeable (trainer.py:105-111, eval.py:29-30).

Round 1: GEMMs / attention go through PyTorch-ROCm library kernels (hipBLASLt,
SDPA) under bf16 autocast; the per-block CLS-row importance tap is computed
from the block's own q/k (no duplicate QKV GEMM, no [B,H,T,T] map).  Parity of
the ViT arithmetic itself is UNPINNED by the reference (it has no tests and
timm is absent) -- see DESIGN.md.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ..losses._ops import get_ops, library_fallback
from .linear import BasdLinear, _LinearFn, fused_mlp, fused_mlp_ok


class DropPath(nn.Module):
    def __init__(self, p: float = 0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        return x * self.scaled_mask(x)

    def scaled_mask(self, x):
        """per-sample keep mask already divided by the keep probability (timm drop_path, scale_by_keep)"""
        keep = 1.0 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return mask / keep

    def add_to(self, residual, x, mask=None):
        """residual + drop_path(x) in one elementwise kernel (addcmul) instead of mul, div and add;
        ``mask``: a pre-drawn scaled keep mask (the model draws all of a forward's masks in one launch)"""
        if self.p == 0.0 or not self.training:
            return residual + x
        return torch.addcmul(residual, x, self.scaled_mask(x) if mask is None else mask)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ops = get_ops()
        y, mean, rstd = ops.layernorm_fwd(x, weight.detach().float(), bias.detach().float(), eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.weight, ctx.bias = weight, bias
        return y

    @staticmethod
    def backward(ctx, g):
        ops = get_ops()
        x, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        g = g.to(torch.bfloat16)
        gw = gb = None
        if weight.requires_grad:
            sink_w, sink_b = getattr(weight, "_basd_grad", None), getattr(bias, "_basd_grad", None)
            if sink_w is not None and sink_b is not None:      # straight into the flat gradient buffer
                dx = ops.layernorm_bwd(g, x, weight.detach().float(), mean, rstd, sink_w, sink_b)
                for p in (weight, bias):
                    ready = getattr(p, "_basd_ready", None)
                    if ready is not None:
                        ready()
            else:
                gw = torch.zeros_like(weight, dtype=torch.float32)
                gb = torch.zeros_like(bias, dtype=torch.float32)
                dx = ops.layernorm_bwd(g, x, weight.detach().float(), mean, rstd, gw, gb)
                gw, gb = gw.to(weight.dtype), gb.to(bias.dtype)
        else:
            dx = ops.layernorm_bwd(g, x, weight.detach().float(), mean, rstd, None, None)
        return dx, gw, gb, None


class _AddLayerNormFn(torch.autograd.Function):
    """One residual step of a TRAINED pre-norm block: s = residual + drop_path(branch), y = LayerNorm(s), both returned.
    Forward: one kernel (``basd_add_layernorm_fwd_bf16`` with the per-sample stochastic-depth scale); backward: one
    kernel that adds the gradient arriving through the residual connection to the LayerNorm backward and also emits the
    (scaled) gradient of the branch -- instead of addcmul + LayerNorm forward, and LayerNorm backward + add (+ mul)."""

    @staticmethod
    def forward(ctx, branch, residual, scale, weight, bias, eps):
        s, y, mean, rstd = get_ops().add_layernorm_fwd(branch, residual, weight.detach().float(), bias.detach().float(),
                                                       eps, row_scale=scale, want_stats=True)
        ctx.save_for_backward(s, mean, rstd)
        ctx.scale, ctx.weight, ctx.bias = scale, weight, bias
        return s, y

    @staticmethod
    def backward(ctx, g_s, g_y):
        ops = get_ops()
        s, mean, rstd = ctx.saved_tensors
        weight, bias, scale = ctx.weight, ctx.bias, ctx.scale
        dy = torch.zeros_like(s) if g_y is None else g_y.to(torch.bfloat16)
        gw = gb = None
        sink_w = sink_b = None
        if weight.requires_grad:
            sink_w, sink_b = getattr(weight, "_basd_grad", None), getattr(bias, "_basd_grad", None)
            if sink_w is None or sink_b is None:
                gw = torch.zeros_like(weight, dtype=torch.float32)
                gb = torch.zeros_like(bias, dtype=torch.float32)
                sink_w, sink_b = gw, gb
        out = ops.layernorm_bwd(dy, s, weight.detach().float(), mean, rstd, sink_w, sink_b, dres=g_s, row_scale=scale,
                                want_branch=scale is not None)
        d_res, d_branch = out if scale is not None else (out, out)
        if weight.requires_grad and gw is None:
            for p in (weight, bias):
                ready = getattr(p, "_basd_ready", None)
                if ready is not None:
                    ready()
        if gw is not None:
            gw, gb = gw.to(weight.dtype), gb.to(bias.dtype)
        return d_branch, d_res, None, gw, gb, None


class MixedLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters) running on the fused HIP kernel for bf16 activations: bf16 in,
    bf16 out, fp32 gamma / beta / statistics.  Equal to torch's autocast behaviour (fp32 layer_norm)
    followed by the bf16 rounding the next Linear applies, without the up/down-cast copies and with a
    one-kernel backward.  Other inputs (fp32 activations, CPU) take the stock path."""

    def forward(self, x):
        pre = getattr(x, "_basd_prenorm", None)
        if pre is not None and pre[0] is self:
            return pre[1]               # already produced by the fused residual-add + norm of the previous step
        if (x.dtype == torch.bfloat16 and get_ops().handles(x) and self.elementwise_affine
                and get_ops().layernorm_supported(x.shape[-1])):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)

    def fused_add(self, x, residual):
        """(x + residual, LayerNorm(x + residual)) in one kernel; inference only."""
        return get_ops().add_layernorm_fwd(x, residual, self.weight.detach().float(), self.bias.detach().float(), self.eps)


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _PackedFlashAttention(torch.autograd.Function):
    """softmax(QK^T / sqrt(hd)) V on the packed projection [B, T, 3 * H * hd] with a packed gradient.

    Through plain autograd the three gradients of the library's flash-attention backward are stacked
    (unbind backward) and then copied once more into the [B, T, 3, H, hd] order the qkv Linear wants:
    two extra passes over the projection gradient per block.  Here the same library kernels are called
    directly (aten::_scaled_dot_product_flash_attention[_backward]) and their three outputs, which are
    [B, T, H, hd]-contiguous, are interleaved once."""

    @staticmethod
    def forward(ctx, qkv_flat, heads, head_dim):
        b, t, _ = qkv_flat.shape
        qkv = qkv_flat.view(b, t, 3, heads, head_dim)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))              # [B, H, T, hd] views
        res = torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False)
        out, lse, cum_q, cum_k, max_q, max_k, seed, offset = res[:8]
        ctx.save_for_backward(qkv_flat, out, lse, cum_q, cum_k, seed, offset)
        ctx.dims = (heads, head_dim, max_q, max_k)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv_flat, out, lse, cum_q, cum_k, seed, offset = ctx.saved_tensors
        heads, head_dim, max_q, max_k = ctx.dims
        b, t, _ = qkv_flat.shape
        qkv = qkv_flat.view(b, t, 3, heads, head_dim)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        dq, dk, dv = torch.ops.aten._scaled_dot_product_flash_attention_backward(
            g, q, k, v, out, lse, cum_q, cum_k, max_q, max_k, 0.0, False, seed, offset)
        dqkv = torch.stack((dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)), dim=2)   # [B, T, 3, H, hd]
        return dqkv.view(b, t, 3 * heads * head_dim), None, None


class _FusedAttention(torch.autograd.Function):
    """softmax(Q K^T / sqrt(hd)) V of a TRAINED block on the hand-written kernels: forward = the fused attention
    kernel the frozen teacher uses, with the log-sum-exp written out; backward = ``basd_attention_bwd_bf16`` (P
    recomputed from Q, K and the LSE, gradient delivered already packed as [B, T, 3 * H * hd])."""

    @staticmethod
    def forward(ctx, qkv_flat, heads, head_dim, scale):
        out, _, lse = get_ops().attention_fwd(qkv_flat, heads, head_dim, scale, want_lse=True)
        ctx.save_for_backward(qkv_flat, out, lse)
        ctx.dims = (heads, head_dim, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv_flat, out, lse = ctx.saved_tensors
        heads, head_dim, scale = ctx.dims
        return get_ops().attention_bwd(qkv_flat, out, g, lse, heads, head_dim, scale), None, None, None


_FUSED_TEACHER_ATTENTION = os.environ.get("BASD_FUSED_ATTN", "1") == "1"     # 0: library SDPA + separate tap (A/B runs)
_packed_attention_ok = True      # cleared on the first failure of the direct library call (other torch builds)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = BasdLinear(dim, dim * 3, bias=qkv_bias)
        self.proj = BasdLinear(dim, dim)
        self.tap: dict | None = None       # set by the teacher tap: {"has_cls": bool, "out": tensor}

    def forward(self, x):
        b, t, c = x.shape
        qkv_flat = self.qkv(x)
        if (_FUSED_TEACHER_ATTENTION and not torch.is_grad_enabled() and qkv_flat.dtype == torch.bfloat16
                and get_ops().handles(qkv_flat) and get_ops().attention_fwd_supported(t, self.head_dim)):
            # frozen block: fused attention straight from the packed projection, the tap as a by-product
            # (csrc/attention.hip)
            # the tap: CLS row of the head-averaged map, or (teachers without a CLS token) that map averaged over the
            # queries -- both by-products of the same kernel
            has_cls = self.tap is not None and bool(self.tap["has_cls"])
            want = self.tap is not None and (t >= 2 or not has_cls)
            out_flat, imp = get_ops().attention_fwd(qkv_flat, self.num_heads, self.head_dim, self.scale, want,
                                                    query_mean=want and not has_cls)
            if self.tap is not None:
                if want:
                    self.tap["out"] = imp
                else:
                    q_, k_, _ = qkv_flat.reshape(b, t, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
                    self.tap["out"] = self._importance(q_, k_, self.tap["has_cls"])
            return self.proj(out_flat)
        if (torch.is_grad_enabled() and qkv_flat.requires_grad and qkv_flat.dtype == torch.bfloat16 and self.tap is None
                and get_ops().handles(qkv_flat) and get_ops().attention_fwd_supported(t, self.head_dim)
                and get_ops().attention_bwd_supported(t, self.head_dim)):
            # trained block (the student): hand-written forward (+ LSE) and backward kernels
            return self.proj(_FusedAttention.apply(qkv_flat.contiguous(), self.num_heads, self.head_dim, self.scale))
        qkv = qkv_flat.reshape(b, t, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        on_device = get_ops().handles(qkv_flat)
        if self.tap is not None:
            ops = get_ops() if on_device else None
            if (ops is not None and self.tap["has_cls"] and qkv_flat.dtype == torch.bfloat16
                    and ops.cls_importance_supported(t, self.head_dim)):
                # fused tap: streams K once from the packed projection (csrc/attn_tap.hip)
                self.tap["out"] = ops.cls_importance(qkv_flat, self.num_heads, self.head_dim, self.scale)
            else:
                if on_device and self.tap["has_cls"]:
                    library_fallback("attention tap", f"T={t} head_dim={self.head_dim} dtype={qkv_flat.dtype}")
                self.tap["out"] = self._importance(q, k, self.tap["has_cls"])
        if on_device:
            library_fallback("attention", f"T={t} head_dim={self.head_dim} dtype={qkv_flat.dtype} "
                                          f"grad={torch.is_grad_enabled() and qkv_flat.requires_grad}")
        global _packed_attention_ok
        out = None
        if (_packed_attention_ok and qkv_flat.is_cuda and qkv_flat.dtype in (torch.bfloat16, torch.float16)
                and torch.is_grad_enabled() and qkv_flat.requires_grad and qkv_flat.is_contiguous()):
            try:
                out = _PackedFlashAttention.apply(qkv_flat, self.num_heads, self.head_dim)
            except (RuntimeError, AttributeError, TypeError):
                _packed_attention_ok = False
        if out is None:
            out = F.scaled_dot_product_attention(q, k, v)
        return self.proj(out.transpose(1, 2).reshape(b, t, c))

    def _importance(self, q, k, has_cls: bool):
        """Head-averaged token importance [B, N]: the only part of softmax(QK^T/sqrt(hd)) the
        loss reads (src/losses/relational.py:22-27; teacher.py:33-37 builds the full map)."""
        if has_cls:
            # q_cls k^T in the activation dtype (bf16 in, fp32 accumulate) like the reference's
            # autocast matmul (teacher.py:36), softmax in fp32; no fp32 copy of K
            logits = (q[:, :, :1] @ k.transpose(-2, -1)).float() * self.scale               # [B,H,1,T]
            return logits.softmax(dim=-1)[:, :, 0, 1:].mean(dim=1)
        attn = ((q.float() @ k.float().transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return attn.mean(dim=(1, 2))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = BasdLinear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = BasdLinear(hidden, dim)

    def forward(self, x):
        if self.fc1.fused_inference_ok(x):        # frozen block: exact-erf GELU in the epilogue of the fc1 GEMM
            return self.fc2(get_ops().gemm_bf16(x, self.fc1.weight, self.fc1.bias, gelu=True))
        if fused_mlp_ok(x, self.fc1, self.fc2):   # trained block: GELU forward / backward inside the GEMM epilogues
            return fused_mlp(x, self.fc1, self.fc2)
        if get_ops().handles(x):
            library_fallback("MLP GELU", f"fc1 {self.fc1.in_features}->{self.fc1.out_features} dtype={x.dtype}")
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, drop_path=0.0, init_values=None, eps=1e-6):
        super().__init__()
        self.norm1 = MixedLayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.ls1 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path)
        self.norm2 = MixedLayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path)

        self._next_norm = []            # [norm that consumes this block's output]; set by the model, not a submodule
        self.fuse_training = True       # cleared under activation checkpointing (tensor attributes do not survive it)

    def _fused_inference(self, x) -> bool:
        """frozen pre-norm block on bf16 activations: the residual adds fuse into the following norms"""
        return (not torch.is_grad_enabled() and x.dtype == torch.bfloat16 and get_ops().handles(x)
                and isinstance(self.ls1, nn.Identity) and isinstance(self.ls2, nn.Identity)
                and (not self.training or (self.drop_path1.p == 0.0 and self.drop_path2.p == 0.0))
                and self.norm2.elementwise_affine and get_ops().layernorm_supported(x.shape[-1]))

    def _fused_training(self, x) -> bool:
        """trained pre-norm block on bf16 activations: residual add (+ stochastic depth) fused with the next norm"""
        return (torch.is_grad_enabled() and self.training and x.dtype == torch.bfloat16 and x.requires_grad
                and get_ops().handles(x) and isinstance(self.ls1, nn.Identity) and isinstance(self.ls2, nn.Identity)
                and self.norm2.elementwise_affine and self.fuse_training and get_ops().layernorm_supported(x.shape[-1]))

    def forward(self, x, dp_masks=None):
        if self._fused_training(x):
            # (bf16 mask, bf16 mask, fp32 per-sample scale, fp32 per-sample scale): the model draws all of them at once
            m1, m2, sc1, sc2 = dp_masks if dp_masks is not None else (None, None, None, None)
            a = self.attn(self.norm1(x))
            x, z = _AddLayerNormFn.apply(a, x, sc1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            m = self.mlp(z)
            nxt = self._next_norm[0] if self._next_norm else None
            if nxt is None or not nxt.elementwise_affine:
                return self.drop_path2.add_to(x, m, m2)
            out, zn = _AddLayerNormFn.apply(m, x, sc2, nxt.weight, nxt.bias, nxt.eps)
            out._basd_prenorm = (nxt, zn)
            return out
        if self._fused_inference(x):
            a = self.attn(self.norm1(x))
            x, z = self.norm2.fused_add(x, a)
            m = self.mlp(z)
            nxt = self._next_norm[0] if self._next_norm else None
            if nxt is None or not nxt.elementwise_affine:
                return x + m
            out, zn = nxt.fused_add(x, m)
            out._basd_prenorm = (nxt, zn)
            return out
        m1, m2 = dp_masks[:2] if dp_masks is not None else (None, None)
        x = self.drop_path1.add_to(x, self.ls1(self.attn(self.norm1(x))), m1)
        return self.drop_path2.add_to(x, self.ls2(self.mlp(self.norm2(x))), m2)


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        # non-overlapping patches: the stride-p convolution IS a GEMM on unfolded patches
        # ([B*N, 3*p*p] x [3*p*p, D]).  Going through F.conv2d lands on MIOpen's untuned
        # naive bf16 convolution here (24 ms fwd + 31 ms wrw per call at B=256).
        b, c, h, w = x.shape
        p = self.patch_size
        d, k = self.proj.weight.shape[0], c * p * p
        unfolded = x.reshape(b, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5)
        ops = get_ops()
        trained = torch.is_grad_enabled() and self.proj.weight.requires_grad
        if ops.handles(x) and self._bf16_compute(x) and ops.gemm_supported(d, k) and (
                not trained or ops.wgrad_supported(d, k)) and self.proj.bias is not None:
            # own GEMM (forward) / split-M weight gradient; the unfold and the cast to bf16 are ONE copy kernel
            patches = torch.empty(b, (h // p) * (w // p), k, dtype=torch.bfloat16, device=x.device)
            patches.view(b, h // p, w // p, c, p, p).copy_(unfolded)
            w2 = _matrix_view(self.proj.weight)
            if trained:
                with torch.autocast(device_type=x.device.type, enabled=False):
                    return _LinearFn.apply(patches, w2, self.proj.bias)
            if not torch.is_grad_enabled() and w2.dtype == torch.bfloat16 and self.proj.bias.dtype == torch.bfloat16:
                return ops.gemm_bf16(patches, w2, self.proj.bias)
            library_fallback("patch embedding", f"K={k} D={d}: bf16 compute of fp32 frozen weights")
            return F.linear(patches, w2.to(torch.bfloat16), self.proj.bias.to(torch.bfloat16))
        k_pad = (k + 63) // 64 * 64
        if (ops.handles(x) and not torch.is_grad_enabled() and self.proj.weight.dtype == torch.bfloat16
                and self.proj.bias is not None and self.proj.bias.dtype == torch.bfloat16 and ops.gemm_supported(d, k_pad)):
            # frozen bf16 layer whose patch is not a multiple of 64 values (ViT-H/14: 3 x 14 x 14 = 588): K padded with
            # zero columns on both operands (the padded weight is kept: the layer is frozen)
            patches = torch.zeros(b, (h // p) * (w // p), k_pad, dtype=torch.bfloat16, device=x.device)
            patches[..., :k].view(b, h // p, w // p, c, p, p).copy_(unfolded)
            cache = getattr(self, "_w_pad", None)
            if cache is None or cache[0] != self.proj.weight._version or cache[1].device != x.device:
                w_pad = torch.zeros(d, k_pad, dtype=torch.bfloat16, device=x.device)
                w_pad[:, :k].copy_(self.proj.weight.reshape(d, k))
                cache = (self.proj.weight._version, w_pad)
                self._w_pad = cache
            return ops.gemm_bf16(patches, cache[1], self.proj.bias)
        if (ops.handles(x) and trained and self._bf16_compute(x) and self.proj.bias is not None
                and ops.gemm_supported(d, k_pad) and ops.wgrad_supported(d, k_pad)):
            # trained layer with a short patch (BASELINE c1: 3 x 4 x 4 = 48 values): the same zero padding of K, the
            # weight padded per step inside the autograd function, its gradient cut back to [D, K]
            patches = torch.zeros(b, (h // p) * (w // p), k_pad, dtype=torch.bfloat16, device=x.device)
            patches[..., :k].view(b, h // p, w // p, c, p, p).copy_(unfolded)
            with torch.autocast(device_type=x.device.type, enabled=False):
                return _PaddedPatchFn.apply(patches, self.proj.weight.view(d, k), self.proj.bias)
        if ops.handles(x):
            library_fallback("patch embedding", f"K={k} D={d} dtype={x.dtype}")
        return F.linear(unfolded.reshape(b, -1, k), self.proj.weight.reshape(d, -1), self.proj.bias)

    def _bf16_compute(self, x) -> bool:
        if self.proj.weight.dtype == torch.bfloat16:
            return True                                      # frozen bf16 teacher
        dev = x.device.type
        return torch.is_autocast_enabled(dev) and torch.get_autocast_dtype(dev) == torch.bfloat16


class _PaddedPatchFn(torch.autograd.Function):
    """patches [B, N, K_pad] bf16 (zero beyond K) x weight [D, K] fp32 master + bias: forward on basd_gemm_bf16 with the
    weight zero-padded to K_pad, weight / bias gradient on basd_wgrad_bf16 (the K_pad - K extra columns are dropped).
    The images need no gradient."""

    @staticmethod
    def forward(ctx, patches, weight, bias):
        d, k = weight.shape
        w16 = torch.zeros(d, patches.shape[-1], dtype=torch.bfloat16, device=patches.device)
        w16[:, :k].copy_(weight)
        ctx.save_for_backward(patches)
        ctx.k = k
        return get_ops().gemm_bf16(patches, w16, bias.to(torch.bfloat16))

    @staticmethod
    def backward(ctx, g):
        (patches,) = ctx.saved_tensors
        g16 = g.to(torch.bfloat16).contiguous()
        gw, gb = get_ops().wgrad_bf16(g16.reshape(-1, g16.shape[-1]), patches.reshape(-1, patches.shape[-1]), need_bias=True)
        return None, gw[:, :ctx.k].contiguous(), gb


def _matrix_view(weight: torch.Tensor) -> torch.Tensor:
    """[D, C, p, p] convolution weight as the [D, C p p] matrix of the GEMM; the bf16 image, the gradient slot in the
    flat buffer and the data-parallel hook of the parameter (trainer attributes) follow the view"""
    w2 = weight.view(weight.shape[0], -1)
    for name in ("_basd_bf16", "_basd_grad"):
        t = getattr(weight, name, None)
        if t is not None:
            setattr(w2, name, t.view(weight.shape[0], -1))
    ready = getattr(weight, "_basd_ready", None)
    if ready is not None:
        w2._basd_ready = ready
    return w2


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, drop_path_rate=0.0, init_values=None, class_token=True):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.num_classes = num_classes
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.pos_embed = nn.Parameter(torch.randn(1, n + int(class_token), embed_dim) * 0.02)
        dpr = [drop_path_rate * i / max(depth - 1, 1) for i in range(depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, dpr[i], init_values) for i in range(depth)])
        self.norm = MixedLayerNorm(embed_dim, eps=1e-6)
        for i, blk in enumerate(self.blocks):
            blk._next_norm.append(self.blocks[i + 1].norm1 if i + 1 < depth else self.norm)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.grad_checkpointing = False
        if self.cls_token is not None:
            nn.init.normal_(self.cls_token, std=1e-6)

    def set_grad_checkpointing(self, enable: bool = True):
        self.grad_checkpointing = enable
        for blk in self.blocks:
            blk.fuse_training = not enable

    def forward_features(self, x):
        x = self.patch_embed(x)
        if self.cls_token is not None:
            x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x], dim=1)
        x = x + self.pos_embed.to(x.dtype)
        masks = self._draw_drop_path_masks(x)
        for i, blk in enumerate(self.blocks):
            dp = None if masks is None else (masks[0][2 * i], masks[0][2 * i + 1], masks[1][2 * i], masks[1][2 * i + 1])
            if self.grad_checkpointing and self.training and torch.is_grad_enabled():
                x = checkpoint(blk, x, dp, use_reentrant=False)
            else:
                x = blk(x, dp)
        return self.norm(x)

    def _draw_drop_path_masks(self, x):
        """All 2 * depth stochastic-depth masks of one forward from ONE bernoulli launch (timm draws one per
        DropPath call: 3 tiny kernels x 24), already divided by the keep probability: ([2 * depth, B, 1, 1] in the
        activation dtype, [2 * depth, B] fp32)."""
        if not self.training:
            return None
        ps = [p for blk in self.blocks for p in (blk.drop_path1.p, blk.drop_path2.p)]
        if not any(p > 0.0 for p in ps):
            return None
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != x.device or keep.shape[0] != len(ps):
            keep = torch.tensor([1.0 - p for p in ps], device=x.device, dtype=torch.float32).view(-1, 1)
            self._dp_keep = keep
        m = torch.bernoulli(keep.expand(-1, x.shape[0])) / keep          # fp32 [2 * depth, B]: the fused kernels' scales
        return m.to(x.dtype).view(len(ps), x.shape[0], 1, 1), m

    def forward(self, x):
        x = self.forward_features(x)
        x = x[:, 0] if self.cls_token is not None else x.mean(dim=1)
        return self.head(x)


# name -> (embed_dim, depth, heads, patch, layerscale init)
VIT_PRESETS = {
    "deit_tiny_patch16_224": (192, 12, 3, 16, None),
    "deit_small_patch16_224": (384, 12, 6, 16, None),
    "deit_base_patch16_224": (768, 12, 12, 16, None),
    "vit_tiny_patch16_224": (192, 12, 3, 16, None),
    "vit_small_patch16_224": (384, 12, 6, 16, None),
    "vit_base_patch16_224": (768, 12, 12, 16, None),
    "vit_large_patch16_224": (1024, 24, 16, 16, None),
    "vit_huge_patch14_224": (1280, 32, 16, 14, None),
    "dinov2_vits14": (384, 12, 6, 14, 1.0),
    "dinov2_vitb14": (768, 12, 12, 14, 1.0),
    "dinov2_vitl14": (1024, 24, 16, 14, 1.0),
}


def create_vit(name: str, *, num_classes: int, img_size: int, drop_path_rate: float = 0.0,
               patch_size: int | None = None, **overrides) -> VisionTransformer:
    """timm.create_model stand-in for the presets above; ``overrides`` are the reference's
    ``model.arch_overrides`` keys (embed_dim, depth, num_heads, mlp_ratio; src/train.py:57-66)."""
    if name not in VIT_PRESETS:
        raise ValueError(f"unknown ViT preset {name!r}; known: {sorted(VIT_PRESETS)}")
    dim, depth, heads, patch, ls = VIT_PRESETS[name]
    kw = dict(embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4.0)
    kw.update({k: v for k, v in overrides.items() if k in ("embed_dim", "depth", "num_heads", "mlp_ratio")})
    return VisionTransformer(img_size=img_size, patch_size=patch_size or patch, num_classes=num_classes,
                             drop_path_rate=drop_path_rate, init_values=ls, **kw)
