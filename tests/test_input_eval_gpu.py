"""Input side and evaluation on the device: MixUp / CutMix soft targets (reference src/training/trainer.py:89-92,138),
top-1 / top-5 evaluation and the efficiency probe (src/evaluation/metrics.py:19-97), and one epoch of the product loop
fed by the local dual-view loader (src/data/datasets.py:126-178)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml")


def test_mixup_cutmix_on_device_soft_targets_and_in_place_outputs():
    from basd_amd.training.mixup import mixup_cutmix
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (16,), generator=g).cuda()
    out, out_t = torch.empty_like(x), torch.empty(16, 10, device="cuda")
    seen = set()
    for trial in range(24):
        mixed, tgt = mixup_cutmix(x, y, 10, generator=torch.Generator().manual_seed(trial), out=out, out_targets=out_t)
        assert mixed.data_ptr() == out.data_ptr() and tgt.data_ptr() == out_t.data_ptr()
        torch.testing.assert_close(tgt.sum(1), torch.ones(16, device="cuda"))
        prev = x.roll(1, 0)
        onehot, onehot_prev = torch.nn.functional.one_hot(y, 10).float(), torch.nn.functional.one_hot(y.roll(1, 0), 10).float()
        # lam from the targets of a sample whose partner has another label
        i = int((y != y.roll(1, 0)).nonzero()[0])
        lam = float(tgt[i, y[i]])
        torch.testing.assert_close(tgt, lam * onehot + (1 - lam) * onehot_prev, atol=1e-6, rtol=0)
        same = (mixed == x)
        if bool(torch.allclose(mixed, lam * x + (1 - lam) * prev, atol=1e-5)):
            seen.add("mixup")
        else:
            # CutMix: every pixel is either the sample's own or its partner's, the pasted share is 1 - lam
            pasted = (mixed == prev) & ~same
            assert bool((same | (mixed == prev)).all())
            share = float(pasted[:, 0].float().mean())
            assert abs(share - (1 - lam)) < 0.02, (share, lam)
            seen.add("cutmix")
    assert seen == {"mixup", "cutmix"}


def _student(classes=10):
    from basd_amd.models.vit import create_vit
    torch.manual_seed(0)
    return create_vit("deit_tiny_patch16_224", num_classes=classes, img_size=32, patch_size=4).cuda()


def test_evaluate_model_matches_a_plain_evaluation():
    from basd_amd.evaluation import evaluate_model, measure_efficiency
    model = _student()
    g = torch.Generator().manual_seed(1)
    batches = [{"pixel_values": torch.randn(n, 3, 32, 32, generator=g).cuda(),
                "label": torch.randint(0, 10, (n,), generator=g).cuda()} for n in (32, 32, 7)]
    crit = torch.nn.CrossEntropyLoss()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = evaluate_model(model, batches, crit, num_classes=10)
    model.eval()
    hits1 = hits5 = n = 0
    loss = 0.0
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for b in batches:
            logits = model(b["pixel_values"]).float()
            top = logits.topk(5, dim=1).indices
            hits1 += int((top[:, 0] == b["label"]).sum())
            hits5 += int((top == b["label"][:, None]).any(1).sum())
            loss += float(crit(logits, b["label"])) * len(b["label"])
            n += len(b["label"])
    assert abs(got["val_acc"] - 100.0 * hits1 / n) < 1e-9 and abs(got["val_acc_top5"] - 100.0 * hits5 / n) < 1e-9
    assert abs(got["loss"] - loss / n) < 1e-4 * abs(loss / n)
    # a class subset (ImageNet-R style): labels index the SUBSET's columns; with 4 columns top-5 always hits
    sub_batch = [{"pixel_values": batches[0]["pixel_values"], "label": batches[0]["label"] % 4}]
    sub = evaluate_model(model, sub_batch, crit, num_classes=4, valid_indices=[1, 3, 5, 7])
    assert 0.0 <= sub["val_acc"] <= 100.0 and sub["val_acc_top5"] == 100.0
    eff = measure_efficiency(model, image_size=32, batch_size=16, num_warmup=2, num_batches=5)
    assert all(v == v and v > 0 for v in eff.values()), eff


def test_one_epoch_of_the_product_loop_from_the_local_dual_view_loader(tmp_path):
    """create_dataloaders (clean view normalised with the teacher's statistics, augmented view with the dataset's) ->
    Trainer._train_epoch with MixUp / CutMix on and the captured step -> evaluate_model on the evaluation loader"""
    from basd_amd.config import load_config
    from basd_amd.data import create_dataloaders
    from basd_amd.evaluation import evaluate_model
    from basd_amd.train import build
    rng = np.random.default_rng(0)
    root = tmp_path / "toy"
    root.mkdir()
    yy, xx = np.meshgrid(np.linspace(-1, 1, 40), np.linspace(-1, 1, 48), indexing="ij")
    for split, n in (("train", 96), ("validation", 40)):
        labels = np.arange(n) % 5
        base = 127 + 80 * np.sin(3 * xx[None] + labels[:, None, None]) * np.cos(2 * yy[None])
        imgs = np.clip(base[..., None] + rng.normal(0, 25, size=(n, 40, 48, 3)), 0, 255).astype(np.uint8)
        np.savez(root / f"{split}.npz", images=imgs, labels=labels)
    cfg = load_config(CFG, "basd_cifar100", [f"data.dataset={root}", "data.batch_size=32", "model.drop_path_rate=0.0"])
    assert cfg.model.num_classes == 5
    trainer, _ = build(cfg, device="cuda")
    train, val = create_dataloaders(cfg, teacher_stats=(trainer._teacher.mean, trainer._teacher.std), num_workers=2)
    trainer.optimizer.train()
    trainer.model.train()
    metrics = trainer._train_epoch(train)
    assert trainer._graph is not None, trainer.graph_error
    assert metrics["train_loss"] == metrics["train_loss"] and trainer.optimizer.k == 3
    trainer.optimizer.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res = evaluate_model(trainer.model, val, torch.nn.CrossEntropyLoss(), num_classes=5)
    assert 0.0 <= res["val_acc"] <= 100.0 and res["val_acc_top5"] == 100.0 and res["loss"] == res["loss"]
