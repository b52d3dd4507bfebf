"""Shared driver: run basd_amd's BASDLoss on a fixture and collect what the goldens hold."""
from __future__ import annotations

import types

import torch


def run_basd_loss(shape, inputs, gold, kind, device="cpu", token_dtype=torch.float32):
    from basd_amd.losses import BASDLoss
    crit = torch.nn.CrossEntropyLoss(label_smoothing=1.0 / shape.C)
    mod = BASDLoss(crit, shape.D_s, shape.D_t, shape.L_s, shape.N_s,
                   config=types.SimpleNamespace(num_extraction_points=shape.E),
                   teacher_has_cls_token=shape.has_cls)
    with torch.no_grad():
        mod.layer_selector.proj_s.copy_(gold["proj_s"])
        mod.layer_selector.proj_t.copy_(gold["proj_t"])
        mod.layer_selector.log_temperatures.copy_(gold["log_temperatures"])
    mod = mod.to(device)
    s_tok = {l: t.to(device=device, dtype=token_dtype).requires_grad_(True)
             for l, t in inputs["student_tokens"].items()}
    logits = inputs["logits"].to(device).requires_grad_(True)
    targets = (inputs["targets_hard"] if kind == "hard" else inputs["targets_soft"]).to(device)
    t_tok = {j: t.to(device=device, dtype=token_dtype) for j, t in inputs["teacher_tokens"].items()}
    t_att = {j: t.to(device) for j, t in inputs["teacher_attns"].items()}
    from basd_amd.losses._ops import get_ops
    get_ops().status_word(device).zero_()
    loss = mod(logits, targets, s_tok, t_tok, t_att)
    loss.backward()
    get_ops().check_status()          # no kernel may have flagged non-convergence / non-finite values / rank 0
    sel = mod.layer_selector
    res = {
        "loss": loss.detach().cpu(), "ce": mod.last_terms["ce"].cpu(), "geo": mod.last_terms["geo"].cpu(),
        "weights": sel.last_weights.cpu(), "pre_softmax": sel.last_pre_softmax.cpu(),
        "ranks": torch.tensor([sel.subspace_ranks[j] for j in sorted(sel.subspace_ranks)]),
        "grad_logits": logits.grad.cpu(), "grad_log_temperatures": sel.log_temperatures.grad.cpu(),
    }
    for l, t in s_tok.items():
        res[f"grad_student_{l}"] = t.grad.float().cpu()
    return res


def check_against_golden(gold, res, kind, layers, *, grad_tol, has_temp_grad=True, geo_rtol=2e-5, check_grads=True):
    from tests._golden import rel_l2
    assert res["ranks"].tolist() == gold[f"{kind}/ranks"].tolist()
    torch.testing.assert_close(res["weights"], gold[f"{kind}/weights"], atol=2e-6, rtol=0)
    if gold[f"{kind}/weights"].shape[1] > 1:      # a single teacher layer: the distance is never formed (weight == 1)
        torch.testing.assert_close(res["pre_softmax"], gold[f"{kind}/pre_softmax"], atol=2e-5, rtol=1e-4)
    print("  geo rel err", ((res["geo"] - gold[f"{kind}/geo"]).abs() / gold[f"{kind}/geo"].abs()).tolist(), "weights max abs err", float((res["weights"] - gold[f"{kind}/weights"]).abs().max()))
    torch.testing.assert_close(res["geo"], gold[f"{kind}/geo"], atol=0, rtol=geo_rtol)
    torch.testing.assert_close(res["ce"], gold[f"{kind}/ce"], atol=0, rtol=1e-5)
    torch.testing.assert_close(res["loss"], gold[f"{kind}/loss"], atol=0, rtol=2e-5)
    assert rel_l2(res["grad_logits"], gold[f"{kind}/grad_logits"]) < 2e-5
    if not check_grads:
        return
    if has_temp_grad:
        torch.testing.assert_close(res["grad_log_temperatures"], gold[f"{kind}/grad_log_temperatures"],
                                   atol=1e-7, rtol=5e-4)
    for l in layers:
        want = gold[f"{kind}/grad_student_{l}"]
        got = res[f"grad_student_{l}"][: want.shape[0]]
        err = rel_l2(got, want)
        print(f"  grad_student_{l}: rel-L2 error vs reference {err:.2e}")
        assert err < grad_tol, (l, err)
