"""``basd-train`` driver on the MI355X path (reference ``src/train.py:72-160``).

Keeps ``_apply_fan_in_init`` (:19-32), ``_create_student`` (:35-54) and
``_derive_from_teacher`` (:57-66) semantics.  Dataset streaming, teacher weight
download and the eval suite need the network and are out of scope (SURVEY
section 2): the driver trains on synthetic dual-view batches unless the caller
passes loaders, with teacher weights random-initialised or loaded from
``basd.teacher_weights``.

    python -m basd_amd.train --config configs/config.yaml [--experiment NAME] [key=value ...]
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import torch
import torch.nn as nn

from .config import load_config
from .models.teacher import TeacherModel, estimate_intrinsic_dim, load_teacher, probe_model
from .models.vit import create_vit
from .training.trainer import Trainer


def _apply_fan_in_init(model: nn.Module) -> None:
    """The reference's student initialisation rule (src/train.py:19-32), stated as a table: He-style standard deviations
    sqrt(2 / fan) -- fan-in, truncated normal, for Linear; fan-out (k_h k_w C_out / groups), normal, for Conv2d --
    LayerNorm at (1, 0) and every bias at 0.  Modules are visited in ``model.modules()`` order, so a seeded run draws
    the same random numbers as the reference's."""
    def he_std(fan: int) -> float:
        return math.sqrt(2.0 / fan)

    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, nn.LayerNorm):
                if mod.elementwise_affine:
                    mod.weight.fill_(1.0)
                    mod.bias.zero_()
                continue
            if isinstance(mod, nn.Linear):
                nn.init.trunc_normal_(mod.weight, std=he_std(mod.in_features))
            elif isinstance(mod, nn.Conv2d):
                k_h, k_w = mod.kernel_size
                nn.init.normal_(mod.weight, std=he_std(k_h * k_w * mod.out_channels // mod.groups))
            else:
                continue
            if mod.bias is not None:
                mod.bias.zero_()


def _create_student(model_name: str, *, num_classes: int, drop_path_rate: float, img_size: int,
                    arch_overrides: dict | None = None, patch_size: int | None = None,
                    grad_checkpointing: bool = False, device="cuda") -> nn.Module:
    model = create_vit(model_name, num_classes=num_classes, drop_path_rate=drop_path_rate, img_size=img_size,
                       patch_size=patch_size, **(arch_overrides or {}))
    _apply_fan_in_init(model)
    # the reference turns activation checkpointing on (train.py:53) to fit 24-80 GB parts; with
    # 288 GB of HBM3E it only costs a second student forward, so it is opt-in here
    model.set_grad_checkpointing(grad_checkpointing)
    return model.to(device)


def _derive_from_teacher(teacher: TeacherModel, intrinsic_dim: int) -> dict:
    """student width = the teacher's intrinsic dimension rounded up to whole teacher heads, capped at the teacher width;
    depth / MLP ratio / head size follow the teacher (the rule of reference src/train.py:57-66)"""
    head_dim = teacher.embed_dim // teacher.heads_per_layer[0]
    d_s = min(math.ceil(intrinsic_dim / head_dim) * head_dim, teacher.embed_dim)
    return {"embed_dim": d_s, "depth": teacher.depth, "num_heads": d_s // head_dim, "mlp_ratio": teacher.mlp_ratio}


class SyntheticLoader:
    """Device-resident dual-view batches with the reference's batch contract
    (src/data/datasets.py:152-156): {"clean", "augmented", "label"}."""

    def __init__(self, batch_size, img_size, num_classes, steps, device, seed=1234):
        g = torch.Generator(device="cpu").manual_seed(seed)
        base = torch.randn(batch_size, 3, img_size, img_size, generator=g)
        # low-frequency structure so that teacher tokens are not pure noise (MP rank >= 1)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, img_size), torch.linspace(-1, 1, img_size), indexing="ij")
        phase = torch.rand(batch_size, 3, 1, 1, generator=g) * 6.28
        base = base + 2.0 * torch.sin(3.0 * xx + phase) * torch.cos(2.0 * yy + phase)
        self.clean = base.to(device)
        self.aug = (base + 0.1 * torch.randn(base.shape, generator=g)).to(device)
        self.label = torch.randint(0, num_classes, (batch_size,), generator=g).to(device)
        self.steps = steps

    def __iter__(self):
        for _ in range(self.steps):
            yield {"clean": self.clean, "augmented": self.aug, "label": self.label}

    def __len__(self):
        return self.steps


class SyntheticEvalLoader:
    """Evaluation batches with the reference's eval contract {"pixel_values", "label"} (src/data/datasets.py:97-123)."""

    def __init__(self, batch_size, img_size, num_classes, steps, device, seed=4321):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.batches = [{"pixel_values": torch.randn(batch_size, 3, img_size, img_size, generator=g).to(device),
                         "label": torch.randint(0, num_classes, (batch_size,), generator=g).to(device)}
                        for _ in range(steps)]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def build(config, device="cuda"):
    torch.manual_seed(config.run.seed)
    img_size = config.model.vit.img_size
    patch = config.model.vit.patch_size
    teacher = load_teacher(config.basd.teacher_model_name, img_size=img_size, device=device,
                           weights=config.basd.get("teacher_weights"), seed=config.run.seed,
                           patch_size=config.basd.get("teacher_patch_size"))
    arch = dict(config.model.get("arch_overrides") or {})
    if not arch and teacher.feature_format == "token" and config.basd.get("derive_student", False):
        calib = torch.randn(math.ceil(10 * teacher.embed_dim / (img_size // patch) ** 2), 3, img_size, img_size,
                            device=device)
        arch = _derive_from_teacher(teacher, estimate_intrinsic_dim(teacher, calib))
        config.model.arch_overrides = arch
    student = _create_student(config.model.student_preset, num_classes=config.model.num_classes,
                              drop_path_rate=config.model.drop_path_rate, img_size=img_size,
                              arch_overrides=arch, patch_size=patch, device=device,
                              grad_checkpointing=bool(config.model.get("grad_checkpointing", False)))
    student_info = probe_model(student, img_size)
    trainer = Trainer(student, config, accelerator=None, teacher=teacher, student_info=student_info)
    return trainer, student_info


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(__file__), "configs", "config.yaml"))
    ap.add_argument("--experiment", default=None)
    ap.add_argument("--steps-per-epoch", type=int, default=10)
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    config = load_config(args.config, args.experiment, args.overrides)
    trainer, info = build(config)
    print(f"student_probed embed_dim={info['embed_dim']} depth={info['depth']} num_tokens={info['num_tokens']}")
    from .data import create_dataloaders, is_local_dataset
    local = is_local_dataset(config.data.dataset)
    if local:
        # reference src/train.py:110-113: dual-view training batches (clean view normalised with the TEACHER's
        # statistics) + the evaluation loader, here from a local dataset root
        loader, val = create_dataloaders(config, teacher_stats=(trainer._teacher.mean, trainer._teacher.std))
    elif config.data.dataset != "synthetic" and os.environ.get("BASD_ALLOW_SYNTHETIC", "0") != "1":
        # never train on noise by accident: a dataset name that is not a local directory (the reference configurations
        # name hub datasets) is an error unless synthetic data is asked for explicitly
        raise SystemExit(f"data.dataset={config.data.dataset!r} is not a local directory (hub datasets need the network); "
                         "use data.dataset=synthetic or BASD_ALLOW_SYNTHETIC=1 for synthetic batches")
    else:
        if config.data.dataset != "synthetic":
            print(f"WARNING: data.dataset={config.data.dataset!r} is not available locally; BASD_ALLOW_SYNTHETIC=1: training "
                  "and evaluating on SYNTHETIC noise batches", file=sys.stderr)
        loader = SyntheticLoader(config.data.batch_size, config.model.vit.img_size, config.model.num_classes,
                                 args.steps_per_epoch, trainer.device)
    start_epoch = 0
    if config.checkpoint.get("resume_from"):            # reference src/train.py:147-149
        start_epoch = trainer.load_checkpoint(config.checkpoint.resume_from)
        print(f"resumed_from={config.checkpoint.resume_from} start_epoch={start_epoch}")
    from .evaluation import evaluate_model, run_eval_suite, save_metrics
    if not local:
        val = SyntheticEvalLoader(config.data.batch_size, config.model.vit.img_size, config.model.num_classes, 2,
                                  trainer.device)
    crit = nn.CrossEntropyLoss()
    trainer.train(loader, val, start_epoch=start_epoch,
                  evaluate_fn=lambda m, l: evaluate_model(m, l, crit, num_classes=config.model.num_classes))
    # reference src/train.py:153-160: evaluation weights (optimizer.eval()), eval suite, metrics.json
    trainer.optimizer.eval()
    results = run_eval_suite(trainer.model, config, config_path=args.config, loaders={config.data.dataset: val},
                             efficiency_kwargs=dict(num_warmup=5, num_batches=20))
    out_dir = os.path.join(config.run.output_dir, config.run.name)
    os.makedirs(out_dir, exist_ok=True)
    print(f"metrics_json={save_metrics(results, out_dir)}")


if __name__ == "__main__":
    main()
