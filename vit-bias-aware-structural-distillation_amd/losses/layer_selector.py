"""Grassmannian layer selector on the HIP kernels.

Mirrors the operator surface of the reference ``src/losses/layer_selector.py``
(``marchenko_pastur_rank`` :8-20, ``GrassmannianLayerSelector`` :40-152: same
constructor, buffers ``proj_s``/``proj_t``, parameter ``log_temperatures``,
``temperatures`` property, ``subspace_ranks`` and ``forward`` signature), with
its own implementation underneath (``functional.py``).

Differences that are visible to a caller, all documented in DESIGN.md:
* ranks stay on the device during ``forward`` (no ``.item()``);
  ``subspace_ranks`` is materialised lazily on first access;
* ``forward`` returns the mixed attention as a compact ``[B, 1, 1, T]`` tensor
  (row 0 of a one-head map) that ``geometric_relational_loss`` consumes exactly
  like the reference's full ``[B, H, T, T]`` mix (the loss only ever reads the
  CLS row / the query mean, relational.py:22-27);
* mixing runs in fp32 (the reference mixes in the token dtype, :110).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as BF
from ._ops import get_ops


@torch.no_grad()
def marchenko_pastur_rank(features: torch.Tensor) -> int:
    """MP rank of an [M, D] matrix (reference layer_selector.py:8-20), computed on the GPU.

    Returns a Python int like the reference (this entry is the start-up path,
    src/models/teacher.py:161-177; the train step keeps ranks on the device).
    """
    ops = get_ops()
    m, d = features.shape
    if d > 1024:
        raise BF.BasdShapeError(f"marchenko_pastur_rank: D={d} > 1024 (the device rank count sorts up to 1024 values)")
    # X^T X in fp64: the fused one-pass kernel up to 256 columns, projection-free split-K fp64 MFMA GEMM beyond
    # (ViT-B .. ViT-H teachers: D_t = 768 .. 1280 -> 1024 is covered; the eigen-solve is the blocked one for D > 192)
    eye = torch.eye(d, device=features.device, dtype=torch.float32)
    gram, _ = ops.token_gram(features.contiguous(), eye)
    sigma, _, _ = BF.psd_eig(gram.unsqueeze(0))
    return int(ops.mp_rank(sigma ** 2, m, d, d)[0].item())


class GrassmannianLayerSelector(nn.Module):
    def __init__(self, num_extraction_points: int, student_dim: int, teacher_dim: int):
        super().__init__()
        self.student_dim = student_dim
        proj_s = torch.empty(student_dim, student_dim)
        proj_t = torch.empty(student_dim, teacher_dim)
        nn.init.orthogonal_(proj_s)
        nn.init.orthogonal_(proj_t)
        self.register_buffer("proj_s", proj_s)
        self.register_buffer("proj_t", proj_t)
        self.log_temperatures = nn.Parameter(
            torch.full((num_extraction_points,), math.log(math.exp(1.0) - 1)))
        self._ranks_dev: torch.Tensor | None = None
        self._rank_keys: list[int] = []
        self._frames = None            # teacher half precomputed by precompute_teacher()
        self.last_weights: torch.Tensor | None = None
        self.last_pre_softmax: torch.Tensor | None = None

    @property
    def temperatures(self) -> torch.Tensor:
        return F.softplus(self.log_temperatures)

    @property
    def subspace_ranks(self) -> dict[int, int]:
        """{teacher layer index: k_j}; reading it synchronises with the device."""
        if self._ranks_dev is None:
            return {}
        return dict(zip(self._rank_keys, self._ranks_dev.tolist()))

    def precompute_student(self, student_tokens_per_layer, extraction_indices) -> None:
        """Optional: the student half of the selector statistics ahead of ``forward`` (consumed by the next
        ``mixing_weights`` call for the same token tensors)."""
        toks = [student_tokens_per_layer[l] for l in extraction_indices]
        self._student_pre = ([id(t) for t in toks], BF.student_frames(toks, self.proj_s))

    def teacher_layer_gram(self, tokens):
        """Gram statistics of ONE teacher layer (to be handed to ``precompute_teacher(..., grams=...)``): lets a
        caller launch them layer by layer while the teacher forward is still running."""
        return BF.teacher_gram(tokens.detach(), self.proj_t)

    def precompute_teacher(self, all_teacher_tokens, grams=None, also_wait=()) -> None:
        """Optional: run the teacher half of the selector (Gram statistics, MP ranks, PCA frames) ahead
        of ``forward`` -- e.g. on a side stream while the student forward runs.  Consumed by the next
        ``mixing_weights`` / ``forward`` call.  ``grams``: {layer: teacher_layer_gram(...)} computed earlier;
        ``also_wait``: further CUDA events the consumer has to wait for (e.g. the end of the teacher forward
        when this runs on a different stream)."""
        teacher_indices = sorted(all_teacher_tokens.keys())
        frames = BF.teacher_frames([all_teacher_tokens[j].detach() for j in teacher_indices], self.proj_t,
                                   grams=None if grams is None else [grams[j] for j in teacher_indices])
        if self.proj_t.is_cuda:
            # produced on the caller's current (possibly side) stream: the consumer joins it through this
            # event at the point of use, after its own student statistics have been enqueued
            done = torch.cuda.Event()
            done.record()
            tensors = [t for t in frames.values() if isinstance(t, torch.Tensor)]
            extra = list(also_wait)

            def ready():
                cur = torch.cuda.current_stream()
                cur.wait_event(done)
                for ev in extra:
                    cur.wait_event(ev)
                if not torch.cuda.is_current_stream_capturing():
                    for t in tensors:
                        t.record_stream(cur)
            frames["ready"] = ready
        self._frames = (teacher_indices, frames)

    def mixing_weights(self, student_tokens_per_layer, all_teacher_tokens, extraction_indices):
        teacher_indices = sorted(all_teacher_tokens.keys())
        frames = None
        if self._frames is not None:
            if self._frames[0] == teacher_indices:
                frames = self._frames[1]
            elif "ready" in self._frames[1]:
                self._frames[1]["ready"]()           # unused precomputation: still join its stream
        self._frames = None
        s_list = [student_tokens_per_layer[l] for l in extraction_indices]
        if len(teacher_indices) == 1 and frames is None:
            # single-layer (CNN) teacher: softmax over one layer == 1 exactly, no gradient reaches the student or the
            # temperatures through the selector; only the rank (a reported attribute) is computed
            ranks = BF.teacher_ranks([all_teacher_tokens[teacher_indices[0]].detach()], self.proj_t)
            # (softmax of a single logit: the temperatures receive an exactly-zero gradient, not None, as in the reference)
            weights = torch.softmax(0.0 * self.log_temperatures.float().view(-1, 1), dim=1)
            self._ranks_dev, self._rank_keys = ranks, teacher_indices
            self.last_weights, self.last_pre_softmax = weights.detach(), torch.zeros_like(weights.detach())
            self._student_pre = None
            return weights, teacher_indices
        pre_s, self._student_pre = getattr(self, "_student_pre", None), None
        if pre_s is not None:
            pre_s = pre_s[1] if pre_s[0] == [id(t) for t in s_list] else None
        weights, ranks, pre = BF.selector_weights(
            s_list, [all_teacher_tokens[j] for j in teacher_indices],
            self.proj_s, self.proj_t, self.log_temperatures, frames=frames, pre_student=pre_s)
        self._ranks_dev, self._rank_keys = ranks, teacher_indices
        self.last_weights, self.last_pre_softmax = weights.detach(), pre.detach()
        return weights, teacher_indices

    def forward(self, student_tokens_per_layer, all_teacher_tokens, all_teacher_attns,
                extraction_indices, *, has_cls_token: bool = True):
        weights, teacher_indices = self.mixing_weights(
            student_tokens_per_layer, all_teacher_tokens, extraction_indices)
        mixed = BF.mix_layers(weights, [all_teacher_tokens[j] for j in teacher_indices])
        imp = torch.stack([BF.importance_from_attention(all_teacher_attns[j], has_cls_token)
                           for j in teacher_indices])                       # [L, B, N_t]
        mixed_imp = torch.einsum("el,lbn->ebn", weights, imp)
        mixed_tokens, mixed_attns = {}, {}
        for i, l in enumerate(extraction_indices):
            mixed_tokens[l] = mixed[i]
            row = mixed_imp[i]
            if has_cls_token:       # compact map whose [:, :, 0, 1:] is the importance vector
                row = torch.cat([torch.zeros_like(row[:, :1]), row], dim=1)
            mixed_attns[l] = row.view(row.shape[0], 1, 1, row.shape[1])
        return mixed_tokens, mixed_attns
