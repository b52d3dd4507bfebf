"""Load a golden fixture (tests/golden/<name>.npz) + the inputs it was made from."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle.synth import SHAPES, checksum, make_inputs, token_layers

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str):
    """Returns (shape, inputs dict of torch tensors, golden dict of torch tensors)."""
    shape = SHAPES[name]
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    gold = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if not k.startswith("in/")}
    if any(k.startswith("in/") for k in z.files):
        layers = token_layers(shape.L_s, shape.E)
        inputs = {
            "student_tokens": {l: torch.from_numpy(z[f"in/student_{l}"]) for l in layers},
            "teacher_tokens": {j: torch.from_numpy(z[f"in/teacher_{j}"]) for j in range(shape.L_t)},
            "teacher_attns": {j: torch.from_numpy(z[f"in/attn_{j}"]) for j in range(shape.L_t)},
            "logits": torch.from_numpy(z["in/logits"]),
            "targets_hard": torch.from_numpy(z["in/targets_hard"]),
            "targets_soft": torch.from_numpy(z["in/targets_soft"]),
            "token_layers": layers,
        }
    else:
        inputs = make_inputs(shape, seed=0)
    got = checksum(inputs)
    want = float(gold["checksum"])
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (
        f"fixture {name}: regenerated inputs do not match the golden checksum ({got} vs {want})")
    return shape, inputs, gold


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-300))
