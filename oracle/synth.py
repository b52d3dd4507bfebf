"""Seeded synthetic inputs for the BASD loss path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is imported by the product path; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

Inputs follow SURVEY.md section 8(d): they must carry spectral structure, or the
Marchenko-Pastur rank (reference src/losses/layer_selector.py:8-20) collapses
to 0 and the reference math divides by zero (layer_selector.py:105).

* teacher layer j: planted rank ``r_j = r0 + dr*j`` signal at SNR ``snr`` plus
  N(0,1) noise, so that k_j == r_j;
* student tokens: a prescribed, strictly decaying spectrum so every truncation
  point k_j has a gap (the student is cut at the TEACHER's rank,
  layer_selector.py:96-97);
* attention: softmax(N(0,1)) over keys, shape [B,H,T,T];
* logits N(0,1); hard and soft targets.
"""
from __future__ import annotations

import dataclasses

import torch


@dataclasses.dataclass(frozen=True)
class Shape:
    B: int
    N_s: int
    N_t: int
    D_s: int
    D_t: int
    L_t: int
    H: int
    C: int
    L_s: int = 12
    E: int = 4
    has_cls: bool = True
    r0: int = 3
    dr: int = 2
    snr: float = 3.0
    decay: float = 0.9
    flat_tail: bool = False


SHAPES = {
    # full-rank Procrustes (N_s - 1 >= D_s): gradients are well defined
    "tiny": Shape(B=6, N_s=40, N_t=40, D_s=32, D_t=64, L_t=4, H=2, C=10),
    # teacher grid differs from student grid -> linear resample (combined.py:9-14)
    "tiny_interp": Shape(B=6, N_s=36, N_t=49, D_s=32, D_t=64, L_t=4, H=2, C=10),
    # token teacher without CLS -> query-mean importance (relational.py:25-27)
    "tiny_nocls": Shape(B=6, N_s=40, N_t=40, D_s=32, D_t=64, L_t=4, H=2, C=10, has_cls=False),
    # N_s - 1 < D_s: cross-covariance is rank deficient (values only)
    "tiny_rankdef": Shape(B=8, N_s=16, N_t=16, D_s=32, D_t=64, L_t=4, H=2, C=10),
    # flat noise tail in the student spectrum: the oracle's own noise floor is 1e-3
    "tiny_flat": Shape(B=6, N_s=40, N_t=40, D_s=32, D_t=64, L_t=4, H=2, C=10, flat_tail=True),
    # single teacher layer (CNN-teacher degenerate case, SURVEY 8 "c3")
    "tiny_cnn": Shape(B=6, N_s=36, N_t=9, D_s=32, D_t=96, L_t=1, H=1, C=10, has_cls=False),
    # BASELINE config c1 loss shapes
    "c1": Shape(B=64, N_s=64, N_t=64, D_s=192, D_t=384, L_t=12, H=6, C=100,
                r0=8, dr=4, decay=0.97),
    # BASELINE config c2 loss shapes at reduced batch (full-rank Procrustes: 195 >= 192)
    "c2_b8": Shape(B=8, N_s=196, N_t=196, D_s=192, D_t=768, L_t=12, H=12, C=1000,
                   r0=8, dr=4, decay=0.97),
    # BASELINE config c3 loss shapes at reduced batch: CNN teacher = one "layer" of 7x7 = 49 tokens x 2048
    # channels, uniform attention, resampled 49 -> 196 ("c3 degenerates, exactly")
    "c3_b4": Shape(B=4, N_s=196, N_t=49, D_s=192, D_t=2048, L_t=1, H=1, C=1000, has_cls=False,
                   r0=8, dr=4, decay=0.97),
    # BASELINE config c2 at the BENCHMARKED batch (256 per GPU): pins the large-batch kernel variants
    # (jacobi_blk, 1024-matrix pchol / trinv, aligned fp64 GEMM) against the reference itself
    "c2_b256": Shape(B=256, N_s=196, N_t=196, D_s=192, D_t=768, L_t=12, H=12, C=1000,
                     r0=8, dr=4, decay=0.97),
    # BASELINE config c4 loss shapes at reduced batch: DeiT-S student (D_s = 384), ViT-L/16 teacher
    "c4_b8": Shape(B=8, N_s=196, N_t=196, D_s=384, D_t=1024, L_t=24, H=16, C=1000,
                   r0=8, dr=4, decay=0.985),
    # BASELINE config c5 loss shapes at reduced batch: ViT-B student (D_s = 768), ViT-H/14 teacher
    # (256 patch tokens, T = 257 -> resampled to 196)
    "c5_b4": Shape(B=4, N_s=196, N_t=256, D_s=768, D_t=1280, L_t=32, H=16, C=1000,
                   r0=8, dr=4, decay=0.99),
}


def token_layers(L_s: int, E: int) -> list[int]:
    """Student extraction points, reference src/losses/combined.py:34-40."""
    if E == 1:
        return [L_s - 1]
    return [round(i * (L_s - 1) / (E - 1)) for i in range(E)]


def make_inputs(shape: Shape, seed: int = 0) -> dict:
    g = torch.Generator().manual_seed(seed)
    B, N_s, N_t, D_s, D_t = shape.B, shape.N_s, shape.N_t, shape.D_s, shape.D_t

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32)

    teacher_tokens = {}
    M_t = B * N_t
    for j in range(shape.L_t):
        r = shape.r0 + shape.dr * j
        sig = randn(M_t, r) @ randn(r, D_t) / (r ** 0.5)
        teacher_tokens[j] = (shape.snr * sig + randn(M_t, D_t)).reshape(B, N_t, D_t)

    layers = token_layers(shape.L_s, shape.E)
    student_tokens = {}
    M_s = B * N_s
    for i, l in enumerate(layers):
        q_left, _ = torch.linalg.qr(randn(M_s, D_s))
        q_right, _ = torch.linalg.qr(randn(D_s, D_s))
        if shape.flat_tail:
            r = shape.r0 + shape.dr * (shape.L_t // 2)
            # noise-like tail: close but not equal singular values (exactly equal
            # ones make the reference's svd backward divide by zero)
            spec = 0.7 + 0.6 * torch.rand(D_s, generator=g)
            spec[:r] = 4.0 + torch.arange(r, 0, -1, dtype=torch.float32)
        else:
            spec = shape.decay ** torch.arange(D_s, dtype=torch.float32)
        spec = spec * (M_s ** 0.5) * (1.0 + 0.1 * i)
        z = (q_left * spec) @ q_right.T + 0.3 * randn(1, D_s)
        student_tokens[l] = z.reshape(B, N_s, D_s)

    T = N_t + 1 if shape.has_cls else N_t
    teacher_attns = {
        j: torch.softmax(randn(B, shape.H, T, T), dim=-1) for j in range(shape.L_t)
    }
    logits = randn(B, shape.C)
    hard = torch.randint(0, shape.C, (B,), generator=g)
    soft = torch.softmax(2.0 * randn(B, shape.C), dim=-1)
    return {
        "student_tokens": student_tokens,
        "teacher_tokens": teacher_tokens,
        "teacher_attns": teacher_attns,
        "logits": logits,
        "targets_hard": hard,
        "targets_soft": soft,
        "token_layers": layers,
    }


def checksum(inputs: dict) -> float:
    """Order-dependent fp64 checksum of every input tensor (fixture guard)."""
    acc = 0.0
    k = 1
    for name in ("student_tokens", "teacher_tokens", "teacher_attns"):
        for key in sorted(inputs[name]):
            t = inputs[name][key].double()
            acc += k * float(t.sum()) + 0.5 * k * float((t * t).sum())
            k += 1
    for name in ("logits", "targets_soft"):
        t = inputs[name].double()
        acc += k * float(t.sum()) + 0.5 * k * float((t * t).sum())
        k += 1
    acc += float(inputs["targets_hard"].double().sum())
    return acc
