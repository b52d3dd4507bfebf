"""Dual-view input pipeline over local data (reference src/data/datasets.py): batch contracts, the deterministic parts of
the transforms, the distributional contracts of the random ones, channel statistics."""
import math
import os

import numpy as np
import pytest
import torch


def _make_root(tmp_path, n_train=24, n_val=10, classes=3, hw=(40, 56)):
    rng = np.random.default_rng(0)
    root = tmp_path / "toyset"
    root.mkdir()
    names = np.array([f"class_{c}" for c in range(classes)])
    for split, n in (("train", n_train), ("validation", n_val)):
        imgs = rng.integers(0, 256, size=(n, hw[0], hw[1], 3), dtype=np.uint8)
        imgs[..., 0] //= 2                                    # channel statistics differ per channel
        labels = np.arange(n) % classes
        np.savez(root / f"{split}.npz", images=imgs, labels=labels, class_names=names)
    return str(root)


def _cfg(root, batch=4, img=32, patch=4):
    from basd_amd.config import load_config
    cfg_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml")
    return load_config(cfg_path, None, [f"data.dataset={root}", f"data.batch_size={batch}", f"model.vit.img_size={img}",
                                        f"model.vit.patch_size={patch}"])


def test_dataset_info_channel_stats_and_resolvers(tmp_path):
    from basd_amd.data import dataset_info, get_channel_stats
    root = _make_root(tmp_path)
    info = dataset_info(root)
    assert info["num_classes"] == 3 and info["class_names"] == ("class_0", "class_1", "class_2")
    assert info["train_split"] == "train" and info["eval_split"] == "validation"
    mean, std = get_channel_stats(root)
    with np.load(os.path.join(root, "train.npz")) as z:
        px = z["images"].reshape(-1, 3).astype(np.float64) / 255.0
    assert np.allclose(mean, px.mean(0), atol=1e-9) and np.allclose(std, px.std(0), atol=1e-9)
    cfg = _cfg(root)
    assert cfg.model.num_classes == 3 and abs(cfg.training.label_smoothing - 1.0 / 3.0) < 1e-12
    assert abs(float(cfg.data.eval_crop_ratio) - 32.0 / 40.0) < 1e-12          # resolvers.py: S / (S + 2 p)


def test_dual_view_batches_keep_the_reference_contract(tmp_path):
    from basd_amd.data import create_dataloaders, get_channel_stats
    root = _make_root(tmp_path)
    cfg = _cfg(root)
    t_mean, t_std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    train, val = create_dataloaders(cfg, teacher_stats=(t_mean, t_std), num_workers=0)
    assert len(train) == 24 // 4                                               # drop_last
    batch = next(iter(train))
    assert sorted(batch) == ["augmented", "clean", "label"]
    assert batch["clean"].shape == batch["augmented"].shape == (4, 3, 32, 32)
    assert batch["clean"].dtype == batch["augmented"].dtype == torch.float32 and batch["label"].dtype == torch.int64
    # the clean view is the EVALUATION transform with the teacher's statistics: resize to round(32 / 0.8) = 40 (the
    # images' shorter side: untouched), centre crop, scale, normalise
    from basd_amd.data.datasets import LocalImageSplit
    split = LocalImageSplit(root, "train")
    found = 0
    for i in range(len(split)):
        img = split.image(i)
        want = (img[:, 4:36, 12:44].float() / 255.0 - torch.tensor(t_mean).view(3, 1, 1)) / torch.tensor(t_std).view(3, 1, 1)
        found += int(any(torch.allclose(want, c, atol=1e-6) for c in batch["clean"]))
    assert found == 4
    # two passes over the loader: same clean views per image, different shuffles / augmentations are allowed but every
    # augmented view is finite and normalised with the DATASET statistics (not the teacher's)
    mean, std = get_channel_stats(root)
    aug = torch.cat([b["augmented"] for b in train])
    assert bool(torch.isfinite(aug).all())
    lo = (0.0 - max(mean)) / min(std) - 1e-4
    hi = (1.0 - min(mean)) / min(std) + 1e-4
    assert float(aug.min()) >= lo and float(aug.max()) <= hi
    ev = next(iter(val))
    assert sorted(ev) == ["label", "pixel_values"] and ev["pixel_values"].shape == (4, 3, 32, 32)
    assert [int(b["label"].numel()) for b in val] == [4, 4, 2]                 # evaluation keeps the ragged last batch


def test_augmentation_rng_is_reproducible_and_per_sample(tmp_path):
    from basd_amd.data import create_dataloaders
    root = _make_root(tmp_path)
    cfg = _cfg(root)
    stats = ((0.5, 0.5, 0.5), (0.25, 0.25, 0.25))
    a = torch.cat([b["augmented"] for b in create_dataloaders(cfg, teacher_stats=stats, num_workers=0)[0]])
    b = torch.cat([b["augmented"] for b in create_dataloaders(cfg, teacher_stats=stats, num_workers=0)[0]])
    assert torch.equal(a, b)
    assert not torch.equal(a[0], a[1])


def test_epochs_draw_fresh_augmentations_through_persistent_workers(tmp_path):
    """the epoch must reach the worker processes (persistent workers keep their pickled copy of the dataset):
    two consecutive epochs give different 'augmented' views and identical 'clean' views for the same image; the same
    epoch number replays the same views"""
    from basd_amd.data import create_dataloaders
    root = _make_root(tmp_path)
    cfg = _cfg(root)
    stats = ((0.5, 0.5, 0.5), (0.25, 0.25, 0.25))
    for workers in (0, 2):
        train, _ = create_dataloaders(cfg, teacher_stats=stats, num_workers=workers)
        assert train.dataset.transform.epoch == 0

        def epoch_views(e):
            train.dataset.set_epoch(e)
            views = {}
            for b in train:
                for c, a in zip(b["clean"], b["augmented"]):
                    views[c.numpy().tobytes()] = a.clone()      # the clean view identifies the image
            return views

        e0, e1, e0_again = epoch_views(0), epoch_views(1), epoch_views(0)
        assert set(e0) == set(e1) == set(e0_again) and len(e0) == 24      # same clean views (no two images coincide)
        assert sum(int(not torch.equal(e0[k], e1[k])) for k in e0) >= 20  # fresh crops / flips / ops in the next epoch
        assert all(torch.equal(e0[k], e0_again[k]) for k in e0)           # (seed, epoch, index) is reproducible
        del train


def test_trainer_advances_the_loader_epoch():
    """Trainer.train tells the dataset which epoch it is in (datasets with a set_epoch hook)"""
    import types
    from basd_amd.training.trainer import Trainer
    seen = []

    class _DS:
        def set_epoch(self, e):
            seen.append(e)

    loader = types.SimpleNamespace(dataset=_DS())
    fake = types.SimpleNamespace(
        config=types.SimpleNamespace(training=types.SimpleNamespace(num_epochs=3)),
        optimizer=types.SimpleNamespace(train=lambda: None, eval=lambda: None), model=types.SimpleNamespace(train=lambda: None),
        _train_epoch=lambda l: {"train_loss": 0.0}, metrics_history={"train_loss": []}, best_val_acc=0.0,
        save_checkpoint=lambda *a: None, save_weights=lambda *a: None)
    fake.metrics_history = __import__("collections").defaultdict(list)
    Trainer.train(fake, loader, None, start_epoch=1)
    assert seen == [1, 2]


def test_transform_pieces_follow_the_torchvision_definitions():
    from basd_amd.data import transforms as T
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8)
    assert T.resize(img, 30).shape == (3, 30, 45) and T.resize(img.transpose(1, 2), 30).shape == (3, 45, 30)
    assert T.center_crop(img, 32).shape == (3, 32, 32) and torch.equal(T.center_crop(img, 32), img[:, 14:46, 29:61])
    assert T.center_crop(img[:, :20], 32).shape == (3, 32, 32)                  # zero padding like torchvision
    # RandomResizedCrop parameters: inside the image, area within scale, aspect within ratio (up to rounding)
    areas, aspects = [], []
    for _ in range(300):
        top, left, h, w = T.random_resized_crop_params(60, 90, g)
        assert 0 <= top and top + h <= 60 and 0 <= left and left + w <= 90
        areas.append(h * w / 5400.0)
        aspects.append(w / h)
    assert min(areas) >= 0.08 - 0.02 and max(areas) <= 1.0 and 0.25 < float(np.mean(areas)) < 0.75
    assert min(aspects) >= 0.75 - 0.05 and max(aspects) <= 4.0 / 3.0 + 0.05
    assert T.random_resized_crop(img, 32, g).shape == (3, 32, 32)
    # colour operations: identity at factor 1 / the documented end points
    for f in (T.adjust_brightness, T.adjust_saturation, T.adjust_contrast, T.adjust_sharpness):
        assert torch.equal(f(img, 1.0), img)
    assert int(T.adjust_brightness(img, 0.0).max()) == 0
    sat0 = T.adjust_saturation(img, 0.0).float()
    assert float((sat0[0] - sat0[1]).abs().max()) <= 1 and float((sat0[1] - sat0[2]).abs().max()) <= 1   # grey
    assert torch.equal(T.posterize(img, 8), img) and int(T.posterize(img, 2).unique().numel()) <= 4
    assert torch.equal(T.solarize(img, 256.0), img) and torch.equal(T.solarize(img, 0.0), 255 - img)
    ac = T.autocontrast((img // 2) + 20)
    assert int(ac.amin()) == 0 and int(ac.amax()) == 255
    eq = T.equalize(img)
    assert eq.shape == img.shape and eq.dtype == torch.uint8
    # geometry: rotation by 0 / translation by 0 are the identity, a translation moves content and fills with zeros
    assert torch.equal(T.apply_ta_op(img, "Rotate", 0.0), img) and torch.equal(T.apply_ta_op(img, "TranslateX", 0.0), img)
    moved = T.apply_ta_op(img, "TranslateX", 5.0)
    assert torch.equal(moved[:, :, 5:], img[:, :, :-5]) and int(moved[:, :, :5].max()) == 0
    rot = T.apply_ta_op(img, "Rotate", 90.0)
    assert rot.shape == img.shape
    # TrivialAugmentWide: every operation is reachable, magnitudes stay in the documented ranges
    assert T._ta_magnitude("Rotate", 30) == 135.0 and T._ta_magnitude("TranslateX", 30) == 32.0
    assert abs(T._ta_magnitude("ShearX", 30) - 0.99) < 1e-12 and T._ta_magnitude("Posterize", 30) == 2
    assert T._ta_magnitude("Solarize", 0) == 255.0 and T._ta_magnitude("Solarize", 30) == 0.0
    seen = set()
    for _ in range(400):
        seen.add(T.TA_WIDE_OPS[T._randint(g, len(T.TA_WIDE_OPS))])
        out = T.trivial_augment_wide(img, g)
        assert out.shape == img.shape and out.dtype == torch.uint8
    assert seen == set(T.TA_WIDE_OPS)


def test_image_folder_layout_and_subset_indices(tmp_path):
    from PIL import Image
    from basd_amd.data import dataset_info, get_subset_indices
    from basd_amd.data.datasets import LocalImageSplit
    rng = np.random.default_rng(1)
    for root, classes in (("parent", ["ant", "bee", "cat", "dog"]), ("child", ["dog", "bee"])):
        for split in ("train", "test"):
            for c in classes:
                d = tmp_path / root / split / c
                d.mkdir(parents=True)
                for k in range(2):
                    Image.fromarray(rng.integers(0, 256, size=(20, 24, 3), dtype=np.uint8)).save(d / f"{k}.png")
    parent, child = str(tmp_path / "parent"), str(tmp_path / "child")
    assert dataset_info(parent)["eval_split"] == "test" and dataset_info(parent)["num_classes"] == 4
    assert get_subset_indices(child, parent) == (1, 3)          # bee, dog (sorted class folders) in the parent's list
    assert get_subset_indices(parent, parent) is None
    split = LocalImageSplit(child, "train")
    assert len(split) == 4 and split.image(0).shape == (3, 20, 24) and split.image(0).dtype == torch.uint8
