"""Schedule-Free AdamW.  schedulefree==1.4.1 is not installed, so parity with the
package is unpinned; the update rule is pinned against a hand-written scalar
trace of the published algorithm (CPU, via the emulated kernel) and the fused HIP
kernel against that same trace on the GPU box."""
import math

import pytest
import torch

from tests import _emul


def _scalar_trace(y0, grads, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """One parameter, plain Python floats: AdamWScheduleFree.step, train mode."""
    y, z, v = y0, y0, 0.0
    weight_sum, lr_max = 0.0, -1.0
    out = []
    for k, g in enumerate(grads):
        bc2 = 1 - b2 ** (k + 1)
        lr_max = max(lr, lr_max)
        weight = lr_max ** 2
        weight_sum += weight
        ckp1 = weight / weight_sum
        v = b2 * v + (1 - b2) * g * g
        gn = g / (math.sqrt(v / bc2) + eps) + wd * y
        y = y + ckp1 * (z - y)
        y = y + lr * (b1 * (1 - ckp1) - 1) * gn
        z = z - lr * gn
        out.append((y, z, v))
    return out


def _run(device, provider):
    from basd_amd.losses import _ops
    from basd_amd.training.optim import AdamWScheduleFree, FlatParams
    _ops.set_ops(provider)
    try:
        p = torch.nn.Parameter(torch.tensor([0.7, -1.3, 2.0, 0.01, 5.0], device=device))
        flat = FlatParams([p])
        opt = AdamWScheduleFree(flat, lr=1e-2, weight_decay=0.05)
        opt.train()
        grads = [[0.3, -0.2, 1.5, 0.0, -4.0], [0.1, 0.4, -0.5, 2.0, 0.5], [-0.6, 0.0, 0.25, -1.0, 3.0]]
        seen = []
        for g in grads:
            flat.grad[:5] = torch.tensor(g, device=device)
            opt.step()
            seen.append((p.detach().cpu().clone(), opt.z[:5].cpu().clone(), opt.exp_avg_sq[:5].cpu().clone()))
            opt.zero_grad()
        y_train = p.detach().cpu().clone()
        opt.eval()
        x_eval = p.detach().cpu().clone()
        opt.train()
        return grads, seen, y_train, x_eval, p.detach().cpu().clone(), opt.z[:5].cpu().clone()
    finally:
        _ops.set_ops(None)


def _check(grads, seen, y_train, x_eval, y_back, z):
    y0 = [0.7, -1.3, 2.0, 0.01, 5.0]
    for j in range(5):
        trace = _scalar_trace(y0[j], [g[j] for g in grads], lr=1e-2, wd=0.05)
        for step, (y, zz, v) in enumerate(trace):
            assert float(seen[step][0][j]) == pytest.approx(y, rel=2e-6, abs=1e-7)
            assert float(seen[step][1][j]) == pytest.approx(zz, rel=2e-6, abs=1e-7)
            assert float(seen[step][2][j]) == pytest.approx(v, rel=2e-6, abs=1e-12)
    # eval(): x = y + (1 - 1/beta1)(z - y); train() maps back
    torch.testing.assert_close(x_eval, y_train + (1 - 1 / 0.9) * (z - y_train), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(y_back, y_train, atol=1e-6, rtol=1e-5)


def test_schedule_free_adamw_trace_cpu():
    _check(*_run("cpu", _emul))


@pytest.mark.gpu
def test_schedule_free_adamw_fused_kernel_gpu():
    import basd_amd._native as native
    _check(*_run("cuda", native))
    # a large flat buffer against the torch formulation
    torch.manual_seed(0)
    n = 1_000_003
    y, g, z, v = (torch.randn(n + 1, device="cuda")[:n].clone() for _ in range(4))
    v.abs_()
    ref = [t.clone() for t in (y, g, z, v)]
    kw = dict(lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, ckp1=0.25, bias_correction2=0.3)
    native.sf_adamw_step(y, g, z, v, **kw)
    _emul.sf_adamw_step(ref[0], ref[1], ref[2], ref[3], **kw)
    torch.testing.assert_close(y, ref[0], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(z, ref[2], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(v, ref[3], atol=1e-7, rtol=1e-5)
