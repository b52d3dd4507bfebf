"""CPU baseline for bench.py (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

Times a CPU restatement of the full BASD train step in fp32 on the host cores:
teacher forward with per-block token capture and FULL attention-map capture
(as the reference's hooks do, src/models/teacher.py:27-39,180-216), student
forward with token taps (src/training/trainer.py:16-37), the oracle loss
(oracle/basd_oracle.py, pinned to the reference's own loss code) and its
backward, and a plain Schedule-Free AdamW update.  The reference's own
``src/train.py`` cannot run on CPU (hard-coded ``.cuda()``) and cannot travel to
the GPU box, so this is reported as ``kind: "port"``.
"""
from __future__ import annotations

import os
import time

import torch


def reference_loss_backward(s_model, t_model, x_clean, x_aug, targets, *, proj_s, proj_t, log_temperatures, layers,
                            smoothing):
    """One forward + backward of the BASD step in plain torch on whatever device / dtype the models live on (CPU fp32
    for the checker): frozen teacher with per-block token capture and FULL attention-map capture (the reference's
    hooks, src/models/teacher.py:27-39,180-216), student with token taps (src/training/trainer.py:16-37), the oracle
    loss.  Leaves ``.grad`` on the student parameters and ``log_temperatures``; returns the oracle's output dict."""
    from oracle import basd_oracle as O
    t_tok, t_att, hooks = {}, {}, []
    for i, blk in enumerate(t_model.blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: t_tok.__setitem__(i, o[:, 1:])))

        def attn_hook(m, a, o, i=i):
            xin = a[0]
            b, n, c = xin.shape
            qkv = m.qkv(xin).reshape(b, n, 3, m.num_heads, c // m.num_heads).permute(2, 0, 3, 1, 4)
            t_att[i] = ((qkv[0] @ qkv[1].transpose(-2, -1)) * m.scale).softmax(-1)
        hooks.append(blk.attn.register_forward_hook(attn_hook))
    with torch.no_grad():
        t_model(x_clean)
    for h in hooks:
        h.remove()
    s_tok, hooks = {}, []
    for l in layers:
        hooks.append(s_model.blocks[l].register_forward_hook(lambda m, a, o, l=l: s_tok.__setitem__(l, o[:, 1:])))
    logits = s_model(x_aug)
    for h in hooks:
        h.remove()
    out = O.basd_loss(logits, targets, s_tok, t_tok, t_att, layers=layers, proj_s=proj_s, proj_t=proj_t,
                      log_temperatures=log_temperatures, has_cls=True, smoothing=smoothing)
    out["loss"].backward()
    out["logits"] = logits.detach()
    return out


def cpu_step_images_per_sec(*, student="deit_tiny_patch16_224", teacher="vit_base_patch16_224", img=224,
                            batch=8, num_classes=1000, timed_steps=2, warmup=1, seed=0):
    from basd_amd.models.vit import create_vit
    from oracle import basd_oracle as O
    from oracle.synth import token_layers

    # the GPU box gives one-GPU jobs a 16-core share; os.cpu_count() reports the whole host and
    # oversubscribing OpenMP threads slows the SVD-heavy oracle by more than 10x
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, int(os.environ.get("BASD_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    t_model = create_vit(teacher, num_classes=0, img_size=img).eval()
    for p in t_model.parameters():
        p.requires_grad = False
    s_model = create_vit(student, num_classes=num_classes, img_size=img).train()
    d_s, d_t = s_model.embed_dim, t_model.embed_dim
    proj_s = torch.nn.init.orthogonal_(torch.empty(d_s, d_s))
    proj_t = torch.nn.init.orthogonal_(torch.empty(d_s, d_t))
    log_t = torch.full((4,), 0.5413248546129181, requires_grad=True)
    layers = token_layers(len(s_model.blocks), 4)
    params = list(s_model.parameters()) + [log_t]
    z = [p.detach().clone() for p in params]
    v = [torch.zeros_like(p) for p in params]

    yy, xx = torch.meshgrid(torch.linspace(-1, 1, img), torch.linspace(-1, 1, img), indexing="ij")
    x = torch.randn(batch, 3, img, img) + 2.0 * torch.sin(3.0 * xx) * torch.cos(2.0 * yy)
    labels = torch.randint(0, num_classes, (batch,))

    def step(k):
        out = reference_loss_backward(s_model, t_model, x, x, labels, proj_s=proj_s, proj_t=proj_t,
                                      log_temperatures=log_t, layers=layers, smoothing=1.0 / num_classes)
        lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 0.05
        ckp1 = 1.0 / (k + 1)
        bc2 = 1 - b2 ** (k + 1)
        with torch.no_grad():
            for p, zz, vv in zip(params, z, v):
                g = p.grad
                vv.mul_(b2).addcmul_(g, g, value=1 - b2)
                gn = g / ((vv / bc2).sqrt() + eps) + wd * p
                p.lerp_(zz, ckp1)
                p.add_(gn, alpha=lr * (b1 * (1 - ckp1) - 1))
                zz.sub_(gn, alpha=lr)
                p.grad = None
        return float(out["loss"])

    import sys
    for k in range(warmup):
        t_w = time.perf_counter()
        step(k)
        print(f"[cpu_baseline] warm-up step {k}: {time.perf_counter() - t_w:.1f} s ({threads} threads)", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for k in range(timed_steps):
        t_s = time.perf_counter()
        step(warmup + k)
        print(f"[cpu_baseline] timed step {k}: {time.perf_counter() - t_s:.1f} s", file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / timed_steps
    return {
        "value": batch / dt, "unit": "images/sec", "cores": threads, "kind": "port",
        "sample": f"{student} / {teacher} {img}x{img}, batch {batch}, fp32, {warmup} warm-up + {timed_steps} "
                  f"timed CPU steps ({dt:.2f} s/step): teacher fwd + full attention maps, student fwd/bwd, "
                  "oracle BASD loss fwd/bwd, Schedule-Free AdamW",
    }


if __name__ == "__main__":
    print(cpu_step_images_per_sec())
