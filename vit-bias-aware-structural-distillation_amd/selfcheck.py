"""``__graft_entry__.smoke()``: one small BASD loss step on cuda:0 through the
HIP library, checked against the CPU oracle (the oracle is the checker only)."""
from __future__ import annotations

import types

import torch


def smoke_check(verbose: bool = True) -> None:
    import basd_amd._native as native
    from basd_amd.losses import BASDLoss, _ops
    from oracle import basd_oracle as O
    from oracle.synth import SHAPES, make_inputs

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X); there is no CPU path")
    native.lib()
    _ops.set_ops(None)
    assert _ops.get_ops() is native
    shape = SHAPES["tiny"]
    inputs = make_inputs(shape, seed=0)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mod = BASDLoss(torch.nn.CrossEntropyLoss(label_smoothing=1.0 / shape.C), shape.D_s, shape.D_t, shape.L_s,
                   shape.N_s, config=types.SimpleNamespace(num_extraction_points=shape.E),
                   teacher_has_cls_token=shape.has_cls).to(dev)
    sel = mod.layer_selector
    want = O.basd_loss_and_grads(inputs, proj_s=sel.proj_s.cpu(), proj_t=sel.proj_t.cpu(),
                                 log_temperatures=sel.log_temperatures.detach().cpu(), has_cls=shape.has_cls,
                                 smoothing=1.0 / shape.C)
    s_tok = {l: t.to(dev).requires_grad_(True) for l, t in inputs["student_tokens"].items()}
    logits = inputs["logits"].to(dev).requires_grad_(True)
    loss = mod(logits, inputs["targets_hard"].to(dev), s_tok,
               {j: t.to(dev) for j, t in inputs["teacher_tokens"].items()},
               {j: t.to(dev) for j, t in inputs["teacher_attns"].items()})
    loss.backward()
    torch.cuda.synchronize()
    assert [sel.subspace_ranks[j] for j in range(shape.L_t)] == want["ranks"].tolist()
    torch.testing.assert_close(loss.detach().cpu(), want["loss"], rtol=2e-5, atol=0)
    torch.testing.assert_close(sel.last_weights.cpu(), want["weights"], atol=2e-6, rtol=0)
    for l, t in s_tok.items():
        err = float((t.grad.cpu() - want[f"grad_student_{l}"]).norm() / want[f"grad_student_{l}"].norm())
        assert err < 2e-4, (l, err)
    if verbose:
        print(f"smoke ok: loss {float(loss):.6f} (oracle {float(want['loss']):.6f}), ranks {want['ranks'].tolist()}")
