"""Generate golden vectors by IMPORTING the reference's own loss code on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
is mounted); the GPU box never sees the reference, only the ``.npz`` fixtures
this script writes under ``tests/golden/``.

What is driven (reference files, read-only):
    src/losses/combined.py:18-85       BASDLoss.__init__/forward
    src/losses/layer_selector.py:8-152 marchenko_pastur_rank, selector
    src/losses/relational.py:5-50      geometric_relational_loss

What is recorded per fixture: ranks k_j, pre-softmax logits -d^2/tau, mixing
weights, per-extraction-point Procrustes values, CE, UW-SO total, and the
gradients w.r.t. student tokens, logits and log_temperatures; plus the
orthogonal projection buffers proj_s/proj_t (stored, never regenerated).

Usage:  python oracle/make_golden.py [fixture ...]
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.synth import SHAPES, Shape, checksum, make_inputs  # noqa: E402

REFERENCE = os.environ.get("BASD_REFERENCE", "/root/reference")
# fixtures whose inputs are small enough to be committed in full
BIG_KEEP = 2
STORE_INPUTS = {"tiny", "tiny_interp", "tiny_nocls", "tiny_rankdef", "tiny_flat", "tiny_cnn"}


def _import_reference():
    sys.path.insert(0, REFERENCE)
    import src.losses.combined as ref_combined  # noqa: WPS433
    import src.losses.layer_selector as ref_selector  # noqa: WPS433
    import src.losses.relational as ref_relational  # noqa: WPS433
    return ref_combined, ref_selector, ref_relational


def run_reference(shape: Shape, inputs: dict, targets_kind: str):
    ref_combined, ref_selector, ref_relational = _import_reference()
    torch.manual_seed(0)  # fixes nn.init.orthogonal_ of proj_s / proj_t
    cfg = types.SimpleNamespace(num_extraction_points=shape.E)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=1.0 / shape.C)
    mod = ref_combined.BASDLoss(
        crit, shape.D_s, shape.D_t, shape.L_s, shape.N_s,
        config=cfg, teacher_has_cls_token=shape.has_cls,
    )
    assert mod.token_layers == inputs["token_layers"]
    # distinct temperatures so the tau path is exercised
    with torch.no_grad():
        mod.layer_selector.log_temperatures.add_(
            torch.linspace(-0.3, 0.3, len(mod.token_layers)))

    s_tok = {l: t.clone().requires_grad_(True) for l, t in inputs["student_tokens"].items()}
    logits = inputs["logits"].clone().requires_grad_(True)
    targets = inputs["targets_hard"] if targets_kind == "hard" else inputs["targets_soft"]

    # capture the softmax call inside _mix_for_student_layer (layer_selector.py:108)
    captured = {"pre": [], "w": []}
    real_softmax = ref_selector.F.softmax

    def spy(x, dim):
        out = real_softmax(x, dim=dim)
        captured["pre"].append(x.detach().clone())
        captured["w"].append(out.detach().clone())
        return out

    # capture per-extraction-point Procrustes values (combined.py:69-75)
    geo_vals = []
    real_geo = ref_combined.geometric_relational_loss

    def geo_spy(*a, **k):
        v = real_geo(*a, **k)
        geo_vals.append(v.detach().clone())
        return v

    ref_selector.F = types.SimpleNamespace(softmax=spy, softplus=torch.nn.functional.softplus)
    ref_combined.geometric_relational_loss = geo_spy
    try:
        loss = mod(logits, targets, s_tok, inputs["teacher_tokens"], inputs["teacher_attns"])
        loss.backward()
    finally:
        ref_selector.F = torch.nn.functional
        ref_combined.geometric_relational_loss = real_geo

    with torch.no_grad():
        ce = crit(inputs["logits"], targets)
    sel = mod.layer_selector
    out = {
        "proj_s": sel.proj_s.detach(),
        "proj_t": sel.proj_t.detach(),
        "log_temperatures": sel.log_temperatures.detach(),
        "ranks": torch.tensor([sel.subspace_ranks[j] for j in sorted(sel.subspace_ranks)]),
        "pre_softmax": torch.stack(captured["pre"]),
        "weights": torch.stack(captured["w"]),
        "geo": torch.stack(geo_vals),
        "ce": ce,
        "loss": loss.detach(),
        "grad_logits": logits.grad,
        "grad_log_temperatures": sel.log_temperatures.grad,
    }
    for l in mod.token_layers:
        out[f"grad_student_{l}"] = s_tok[l].grad
    return out


def build(name: str) -> str:
    shape = SHAPES[name]
    inputs = make_inputs(shape, seed=0)
    arrays = {"checksum": np.float64(checksum(inputs))}
    for kind in ("hard", "soft"):
        res = run_reference(shape, inputs, kind)
        for k, v in res.items():
            key = k if k in ("proj_s", "proj_t", "log_temperatures") else f"{kind}/{k}"
            if k.startswith("grad_student_") and name not in STORE_INPUTS:
                # big fixtures: keep the first BIG_KEEP samples + the full-tensor L2 norm
                arrays[key + "_norm"] = np.float64(v.double().norm())
                v = v[:BIG_KEEP]
            arrays[key] = v.numpy()
    if name in STORE_INPUTS:
        for l, t in inputs["student_tokens"].items():
            arrays[f"in/student_{l}"] = t.numpy()
        for j, t in inputs["teacher_tokens"].items():
            arrays[f"in/teacher_{j}"] = t.numpy()
        for j, t in inputs["teacher_attns"].items():
            arrays[f"in/attn_{j}"] = t.numpy()
        arrays["in/logits"] = inputs["logits"].numpy()
        arrays["in/targets_hard"] = inputs["targets_hard"].numpy()
        arrays["in/targets_soft"] = inputs["targets_soft"].numpy()
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: ranks={arrays['hard/ranks'].tolist()} loss={float(arrays['hard/loss']):.6f} "
          f"geo={arrays['hard/geo'].tolist()} -> {path} ({os.path.getsize(path)/1e6:.2f} MB)")
    return path


if __name__ == "__main__":
    names = sys.argv[1:] or list(SHAPES)
    torch.set_num_threads(os.cpu_count() or 1)
    for n in names:
        build(n)
