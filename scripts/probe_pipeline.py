"""Timing probe: what would software-pipelining the frozen teacher ACROSS steps buy?

Today a step is  [teacher fwd(k) || student fwd(k)] -> loss(k) -> backward(k);  the latency-bound kernels of the loss
(the 24 + 48 small eigen-solves, the pivoted Cholesky, ...) leave most of the GPU idle while they run.  Pipelined, the
side stream would compute the teacher forward + statistics of batch k + 1 during loss(k) / backward(k) and the step
would never wait for the teacher.  This script captures both schedules into hipGraphs and times them (the pipelined one
with the held teacher outputs of ONE batch as static inputs and the side branch writing buffers nobody reads: the
timing is representative, the data flow is not wired up).  Result on an MI355X (c2): see DESIGN.md section 9.

    python scripts/probe_pipeline.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import basd_amd._native as native  # noqa: E402
from basd_amd.config import load_config  # noqa: E402
from basd_amd.models.teacher import extract_intermediates  # noqa: E402
from basd_amd.train import SyntheticLoader, build  # noqa: E402
from basd_amd.training.trainer import _extract_student  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    native.lib()
    cfg = load_config(os.path.join(ROOT, "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml"), None,
                      ["data.batch_size=256", "data.dataset=synthetic", "model.student_preset=deit_tiny_patch16_224",
                       "basd.teacher_model_name=vit_base_patch16_224", "model.vit.img_size=224", "model.vit.patch_size=16",
                       "model.grad_checkpointing=false"])
    trainer, _ = build(cfg, device=dev)
    batch = next(iter(SyntheticLoader(256, 224, cfg.model.num_classes, 1, dev, seed=1234)))
    batch2 = next(iter(SyntheticLoader(256, 224, cfg.model.num_classes, 1, dev, seed=99)))
    trainer.optimizer.train()
    trainer.model.train()
    trainer.use_mixup = False
    for _ in range(3):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    sel = trainer.basd_loss.layer_selector
    targets = torch.nn.functional.one_hot(batch["label"], trainer.num_classes).float()

    # ---- held teacher outputs of `batch` (static inputs of the pipelined step)
    with torch.no_grad():
        held_tokens, held_imp = extract_intermediates(trainer._teacher, batch["clean"])
    sel.precompute_teacher(held_tokens)
    held_frames = sel._frames
    held_frames[1].pop("ready", None)
    sel._frames = None
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)

    def student_forward():
        with torch.autocast(device_type="cuda", dtype=trainer.autocast_dtype):
            return _extract_student(trainer.model, batch["augmented"], trainer.basd_loss.token_layers,
                                    layer_paths=trainer._student_layer_paths, has_cls_token=trainer._student_has_cls)

    def piped(split=None):
        """split = j: the side stream pauses after teacher block j until the main stream has finished the loss
        FORWARD (the GPU-filling Jacobi launch of the Procrustes cores), then continues under the backward"""
        trainer.flat.refresh_bf16()
        cur = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record()
        sel._frames = (held_frames[0], dict(held_frames[1]))
        logits, s_tokens = student_forward()
        loss = trainer.basd_loss(logits.float(), targets, s_tokens, held_tokens, held_imp)
        fwd_done = torch.cuda.Event()
        fwd_done.record()
        side.wait_event(start)
        with torch.cuda.stream(side):
            def on_layer(j, _t):
                if split is not None and j == split:
                    side.wait_event(fwd_done)
            tn, _ = extract_intermediates(trainer._teacher, batch2["clean"], on_layer=on_layer)
            sel.precompute_teacher(tn)
            sel._frames = None
        loss.backward()
        cur.wait_stream(side)
        return loss.detach()

    def plain():
        return trainer._forward_backward(batch["clean"], batch["augmented"], targets)[0]

    results = {}
    variants = [("plain", plain), ("pipelined", piped)]
    variants += [(f"pipelined, pause after block {j}", (lambda j=j: piped(j))) for j in (1, 3, 5, 7, 9)]
    for name, fn in variants:
        trainer.reducer.paused = True
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(3):
                fn()
                trainer.flat.zero_grad()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        trainer.flat.zero_grad()
        for _ in range(3):
            g.replay()
            trainer.optimizer.step()
            trainer.optimizer.zero_grad()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 15
        for _ in range(n):
            g.replay()
            trainer.optimizer.step()
            trainer.optimizer.zero_grad()
        torch.cuda.synchronize()
        results[name] = (time.perf_counter() - t0) / n * 1e3
        print(f"{name}: {results[name]:.2f} ms per step (loss {float(out):.4f})", flush=True)
    print(f"pipelining the teacher across steps: {results['plain'] - results['pipelined']:+.2f} ms "
          f"({100 * (results['plain'] / results['pipelined'] - 1):+.1f} % images/s)")
    best = min(results, key=results.get)
    print(f"best schedule: {best} ({results[best]:.2f} ms)")


if __name__ == "__main__":
    main()
