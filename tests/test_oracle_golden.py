"""Pin the CPU oracle (oracle/basd_oracle.py) to golden vectors produced by the
reference's own loss code (oracle/make_golden.py).  CPU only.

Tolerances (fp32, stated per SURVEY 8c): ranks exact; mixing weights atol 2e-6;
pre-softmax rtol 1e-4; per-layer Procrustes / CE / total rtol 1e-5; temperature
grad rtol 2e-4; logits grad rel-L2 1e-5; student-token grads rel-L2 <= 2e-4 on
gapped-spectrum fixtures and <= 5e-3 on the flat-tail fixture (the reference's
own noise floor there is 1e-3).
"""
import pytest
import torch

from oracle import basd_oracle as O
from tests._golden import load, rel_l2

FULL_RANK = ["tiny", "tiny_interp", "tiny_nocls", "tiny_cnn"]


def _run(name, kind, dtype=torch.float32):
    shape, inputs, gold = load(name)
    res = O.basd_loss_and_grads(
        inputs, proj_s=gold["proj_s"], proj_t=gold["proj_t"],
        log_temperatures=gold["log_temperatures"], has_cls=shape.has_cls,
        smoothing=1.0 / shape.C, targets_kind=kind, dtype=dtype)
    return shape, inputs, gold, res


def _check_values(gold, res, kind, *, geo_rtol=1e-5):
    assert res["ranks"].tolist() == gold[f"{kind}/ranks"].tolist()
    torch.testing.assert_close(res["weights"].float(), gold[f"{kind}/weights"], atol=2e-6, rtol=0)
    torch.testing.assert_close(res["pre_softmax"].float(), gold[f"{kind}/pre_softmax"], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(res["geo"].float(), gold[f"{kind}/geo"], atol=0, rtol=geo_rtol)
    torch.testing.assert_close(res["ce"].float(), gold[f"{kind}/ce"], atol=0, rtol=1e-5)
    torch.testing.assert_close(res["loss"].float(), gold[f"{kind}/loss"], atol=0, rtol=1e-5)


@pytest.mark.parametrize("kind", ["hard", "soft"])
@pytest.mark.parametrize("name", FULL_RANK)
def test_oracle_matches_reference_fp32(name, kind):
    shape, inputs, gold, res = _run(name, kind)
    _check_values(gold, res, kind)
    assert rel_l2(res["grad_logits"], gold[f"{kind}/grad_logits"]) < 1e-5
    if shape.L_t > 1:
        torch.testing.assert_close(res["grad_log_temperatures"].float(),
                                   gold[f"{kind}/grad_log_temperatures"], atol=1e-7, rtol=2e-4)
    for l in inputs["token_layers"]:
        assert rel_l2(res[f"grad_student_{l}"], gold[f"{kind}/grad_student_{l}"]) < 2e-4, l


def test_oracle_flat_tail_noise_floor():
    shape, inputs, gold, res = _run("tiny_flat", "hard")
    _check_values(gold, res, "hard")
    for l in inputs["token_layers"]:
        assert rel_l2(res[f"grad_student_{l}"], gold[f"hard/grad_student_{l}"]) < 5e-3, l


def test_oracle_rank_deficient_values_only():
    # N_s - 1 < D_s: the nuclear-norm subgradient the reference returns contains an
    # arbitrary null-space component, so only VALUES are pinned here.
    shape, inputs, gold, res = _run("tiny_rankdef", "hard")
    _check_values(gold, res, "hard")


def test_oracle_fp64_truth_agrees():
    # the fp64 run of the oracle is the "truth" kernels are also compared against
    shape, inputs, gold, res = _run("tiny", "hard", dtype=torch.float64)
    _check_values(gold, res, "hard", geo_rtol=2e-5)
    for l in inputs["token_layers"]:
        assert rel_l2(res[f"grad_student_{l}"], gold[f"hard/grad_student_{l}"]) < 2e-4, l


@pytest.mark.parametrize("name", ["c1", "c2_b8", "c3_b4", "c4_b8", "c5_b4"])
def test_oracle_matches_reference_baseline_shapes(name):
    shape, inputs, gold, res = _run(name, "hard")
    _check_values(gold, res, "hard")
    assert rel_l2(res["grad_logits"], gold["hard/grad_logits"]) < 1e-5
    for l in inputs["token_layers"]:
        g = res[f"grad_student_{l}"]
        n_ref = float(gold[f"hard/grad_student_{l}_norm"])
        assert abs(float(g.double().norm()) - n_ref) <= 2e-4 * n_ref
        if name != "c1":   # c1's committed per-sample slices predate the discovery below; norms are pinned
            # (rank-deficient cores, N_s - 1 < D_s as in c4 / c5: the null-space part of U V^T is annihilated by
            # the token matrices, so the gradients ARE well defined and compared in full)
            keep = gold[f"hard/grad_student_{l}"].shape[0]
            assert rel_l2(g[:keep], gold[f"hard/grad_student_{l}"]) < 5e-4, l
