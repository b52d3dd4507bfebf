"""Evaluation of the trained student: top-1 / top-5 accuracy, inference efficiency, ``metrics.json``.

Operator surface of reference ``src/evaluation/metrics.py`` (``evaluate_model`` :19-55, ``measure_efficiency``
:58-97, ``run_eval_suite`` :100-164, ``save_metrics`` :167-171) with the same result keys.  Differences:

* no torchmetrics: hit counts are accumulated on the device and, under ``torch.distributed``, summed over the ranks
  (what ``MulticlassAccuracy.compute()`` does in the reference);
* the evaluation datasets come from the Hugging Face hub in the reference (network); ``run_eval_suite`` therefore takes
  the loaders from the caller: ``{dataset name: loader | (loader, valid_indices)}``, batches with the reference's eval
  contract ``{"pixel_values", "label"}`` (``src/data/datasets.py:97-123``);
* everything runs on the model's own device (the reference hard-codes ``.cuda()``).
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from typing import Any

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.flop_counter import FlopCounterMode


def _device_of(model: nn.Module) -> torch.device:
    return next(model.parameters()).device


def _sync(device: torch.device) -> None:
    if device.type == "cuda":
        torch.cuda.synchronize(device)


@torch.no_grad()
def evaluate_model(model: nn.Module, data_loader, criterion: nn.Module, *, num_classes: int,
                   valid_indices: list[int] | None = None) -> dict[str, Any]:
    """-> {"val_acc", "val_acc_top5", "loss"} (percent, percent, mean loss); ``valid_indices`` restricts the logits
    to a class subset (ImageNet-R / -A style robustness sets)."""
    model.eval()
    dev = _device_of(model)
    tally = torch.zeros(4, dtype=torch.float64, device=dev)        # hits@1, hits@5, summed loss, samples
    keep = None if valid_indices is None else torch.as_tensor(valid_indices, device=dev)
    for batch in data_loader:
        x = batch["pixel_values"].to(dev, non_blocking=True)
        y = batch["label"].to(dev, non_blocking=True)
        logits = model(x).float()
        if keep is not None:
            logits = logits.index_select(1, keep)
        top = logits.topk(min(5, num_classes, logits.shape[1]), dim=1).indices     # a class subset may have < 5 columns
        hit = top.eq(y.unsqueeze(1))
        tally[0] += hit[:, 0].sum()
        tally[1] += hit.any(dim=1).sum()
        tally[2] += criterion(logits, y).double() * y.numel()
        tally[3] += y.numel()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tally)
    h1, h5, loss, n = tally.tolist()
    n = max(n, 1.0)
    return {"val_acc": 100.0 * h1 / n, "val_acc_top5": 100.0 * h5 / n, "loss": loss / n}


@torch.no_grad()
def measure_efficiency(model: nn.Module, *, image_size: int, in_channels: int = 3, batch_size: int = 64,
                       num_warmup: int = 50, num_batches: int = 200) -> dict[str, float]:
    """Parameter count, forward GFLOPs of one image (``FlopCounterMode``: the evaluation forward of the student runs
    on library ops, which the counter sees) and inference throughput at ``batch_size``."""
    model.eval()
    dev = _device_of(model)
    params = sum(p.numel() for p in model.parameters())
    with FlopCounterMode(display=False) as counter:
        model(torch.randn(1, in_channels, image_size, image_size, device=dev))
    gflops = counter.get_total_flops() / 1e9
    x = torch.randn(batch_size, in_channels, image_size, image_size, device=dev)
    for _ in range(num_warmup):
        model(x)
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(num_batches):
        model(x)
    _sync(dev)
    dt = time.perf_counter() - t0
    return {"param_count": params, "param_count_m": params / 1e6, "gflops": gflops,
            "throughput_img_per_sec": batch_size * num_batches / dt}


def run_eval_suite(model: nn.Module, config, *, config_path: str, loaders: dict, efficiency_kwargs: dict | None = None
                   ) -> dict[str, Any]:
    """Primary dataset + robustness sets + efficiency, in the reference's ``metrics.json`` structure."""
    criterion = nn.CrossEntropyLoss()
    primary, robustness = {}, {}
    for name in [config.data.dataset] + list(config.data.get("eval_datasets") or []):
        if name not in loaders:
            raise KeyError(f"run_eval_suite: no loader for {name!r} (datasets are not fetched here; pass them in)")
        entry = loaders[name]
        loader, subset = entry if isinstance(entry, tuple) else (entry, None)
        classes = len(subset) if subset is not None else config.model.num_classes
        m = evaluate_model(model, loader, criterion, num_classes=classes, valid_indices=subset)
        print(f"eval {name} top1={m['val_acc']:.4f} top5={m['val_acc_top5']:.4f} loss={m['loss']:.6f}")
        if name == config.data.dataset:
            primary = m
        else:
            robustness[name] = m
    eff = measure_efficiency(model, image_size=config.model.vit.img_size, **(efficiency_kwargs or {}))
    print(f"efficiency params_m={eff['param_count_m']:.4f} gflops={eff['gflops']:.4f} "
          f"throughput={eff['throughput_img_per_sec']:.2f} img/s")
    return {"run": {"name": config.run.name, "config": config_path},
            "primary": {"dataset": config.data.dataset, **primary}, "robustness": robustness, "efficiency": eff}


def save_metrics(results: dict[str, Any], output_dir: Path) -> Path:
    path = Path(output_dir) / "metrics.json"
    path.write_text(json.dumps(results, indent=2))
    return path
