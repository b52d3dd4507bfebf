"""BASD loss: CE + mean Procrustes over the extraction points, UW-SO weighted.

Mirrors reference ``src/losses/combined.py`` (``_align_token_count`` :9-14,
``BASDLoss`` :17-85): same constructor and ``forward`` signature, same
``token_layers`` rule, ``layer_selector`` sub-module and state_dict keys.
``all_teacher_attns`` may hold full maps ``[B,H,T,T]`` (reference contract) or
per-layer importance vectors ``[B,N_t]`` (what ``models.teacher`` emits).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as BF
from .layer_selector import GrassmannianLayerSelector


def _align_token_count(tokens: torch.Tensor, target_n: int) -> torch.Tensor:
    """Linear resample along the token axis (reference combined.py:9-14).

    The fused Procrustes prep does this on the fly; the function is kept for callers
    of the reference surface.
    """
    if tokens.shape[1] == target_n:
        return tokens
    r = BF.resample_matrix(tokens.shape[1], target_n, tokens.device, tokens.dtype)
    return torch.matmul(r, tokens)


class _CeUwsoFn(torch.autograd.Function):
    """total = w_ce CE(logits, targets) + w_geo geo with the detached UW-SO weights, one fused forward (which already
    holds d total / d logits) instead of ~25 pointwise / reduction launches; returns (total, ce.detach())."""

    @staticmethod
    def forward(ctx, logits, geo, targets, smoothing):
        out4, dl = BF.get_ops().ce_uwso(logits, targets, smoothing, geo)
        ctx.save_for_backward(dl, out4)
        ctx.mark_non_differentiable(out4)
        return out4[0], out4

    @staticmethod
    def backward(ctx, g, _):
        dl, out4 = ctx.saved_tensors
        return dl * g, out4[3] * g, None, None


class _MixImportanceFn(torch.autograd.Function):
    """mixed_imp[e] = sum_l weights[e, l] imp[l] (reference src/losses/combined.py: the einsum over the teacher layers).
    Autograd's backward of that einsum is ONE [E, B N] x [B N, L] product -- a 50 176-deep reduction the library runs on
    16 x 16 tiles (90 us per step); here it is a batch of B small products summed over the batch."""

    @staticmethod
    def forward(ctx, weights, imp):
        ctx.save_for_backward(weights, imp)
        return torch.einsum("el,lbn->ebn", weights, imp)

    @staticmethod
    def backward(ctx, g):
        weights, imp = ctx.saved_tensors
        g_w = g_imp = None
        if ctx.needs_input_grad[0]:
            g_w = torch.bmm(g.permute(1, 0, 2), imp.permute(1, 2, 0)).sum(dim=0).to(weights.dtype)     # [B, E, L] -> [E, L]
        if ctx.needs_input_grad[1]:
            g_imp = torch.einsum("el,ebn->lbn", weights, g)
        return g_w, g_imp


class BASDLoss(nn.Module):
    def __init__(self, base_criterion: nn.Module, student_dim: int, teacher_dim: int,
                 student_depth: int, num_student_tokens: int, *, config, teacher_has_cls_token: bool):
        super().__init__()
        self.base_criterion = base_criterion
        self.teacher_has_cls_token = teacher_has_cls_token
        self.num_student_tokens = num_student_tokens
        n_pts = config.num_extraction_points
        if n_pts == 1:
            self.token_layers = [student_depth - 1]
        else:
            self.token_layers = [round(i * (student_depth - 1) / (n_pts - 1)) for i in range(n_pts)]
        self.layer_selector = GrassmannianLayerSelector(
            num_extraction_points=len(self.token_layers), student_dim=student_dim, teacher_dim=teacher_dim)
        self.last_terms: dict[str, torch.Tensor] = {}

    def _fused_ce(self, student_output, targets) -> bool:
        """nn.CrossEntropyLoss(label_smoothing) in its default form on device logits: CE, its gradient and the UW-SO
        combination run as one C entry (basd_ce_uwso); any other criterion takes the torch path"""
        c = self.base_criterion
        return (type(c) is nn.CrossEntropyLoss and c.weight is None and c.reduction == "mean"
                and student_output.dim() == 2 and student_output.dtype == torch.float32
                and BF.get_ops().handles(student_output)
                and (targets.dim() == 2 or (targets.dim() == 1 and c.ignore_index < 0)))

    def forward(self, student_output, targets, student_intermediates, all_teacher_tokens, all_teacher_attns):
        fused_ce = self._fused_ce(student_output, targets)
        ce_loss = None if fused_ce else self.base_criterion(student_output, targets)

        sel = self.layer_selector
        weights, teacher_indices = sel.mixing_weights(student_intermediates, all_teacher_tokens, self.token_layers)
        mixed = BF.mix_layers(weights, [all_teacher_tokens[j] for j in teacher_indices])
        imp = torch.stack([BF.importance_from_attention(all_teacher_attns[j], self.teacher_has_cls_token)
                           for j in teacher_indices])
        mixed_imp = _MixImportanceFn.apply(weights, imp)

        students = []
        for layer_idx in self.token_layers:
            s = student_intermediates[layer_idx]
            if s.shape[1] != self.num_student_tokens:
                raise ValueError(f"student layer {layer_idx} has {s.shape[1]} tokens, expected {self.num_student_tokens}")
            students.append(s)
        geo_each = BF.procrustes_all(students, mixed, mixed_imp).mean(dim=1)      # [E]
        geo_loss = geo_each.mean()
        if fused_ce:
            total, out4 = _CeUwsoFn.apply(student_output, geo_loss.float(), targets, float(self.base_criterion.label_smoothing))
            self.last_terms = {"ce": out4[1], "geo": geo_each.detach()}
            return total

        # UW-SO (reference combined.py:78-85): w_i = (1/L_i) / sum_j (1/L_j), detached
        eps = torch.finfo(ce_loss.dtype).eps
        inv = torch.stack([1.0 / ce_loss.detach().clamp(min=eps), 1.0 / geo_loss.detach().float().clamp(min=eps)])
        w = inv / inv.sum()
        self.last_terms = {"ce": ce_loss.detach(), "geo": geo_each.detach()}
        return w[0] * ce_loss + w[1] * geo_loss
