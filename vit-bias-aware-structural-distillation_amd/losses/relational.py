"""Attention-weighted Procrustes loss (reference src/losses/relational.py:5-50).

Same signature as the reference's ``geometric_relational_loss``; the weighted
centring, the cross-covariance, its nuclear norm and the gradient ``U V^T`` run
on the HIP kernels (``functional.procrustes``).
"""
from __future__ import annotations

import torch

from . import functional as BF


def geometric_relational_loss(student_tokens: torch.Tensor, teacher_tokens: torch.Tensor,
                              teacher_attn: torch.Tensor, *, has_cls_token: bool) -> torch.Tensor:
    """student [B,N_s,D_s], teacher [B,N_t,D_t], attn [B,H,T,T] | compact [B,1,1,T] | importance [B,N]."""
    imp = BF.importance_from_attention(teacher_attn, has_cls_token)
    return BF.procrustes(student_tokens, teacher_tokens, imp).mean()
