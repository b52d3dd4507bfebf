"""Dual-view input pipeline over LOCAL data, with the reference's batch contracts (``src/data/datasets.py``):

* training batches ``{"clean": [B,3,S,S] f32, "augmented": [B,3,S,S] f32, "label": [B] i64}`` (:152-156): the clean
  view goes to the frozen teacher (evaluation transform with the TEACHER's mean / std, :146-149), the augmented view to
  the student (:137-144);
* evaluation batches ``{"pixel_values", "label"}`` (:97-123).

The reference streams Hugging Face datasets (``load_dataset(..., trust_remote_code=True)``: network, out of scope
here).  This module reads what is on disk:

* a directory ``root/<split>/<class name>/*.{png,jpg,jpeg,bmp,ppm,webp}`` (decoded with PIL), splits ``train`` and
  ``validation`` | ``val`` | ``test``;
* or one ``.npz`` per split, ``root/<split>.npz`` with ``images`` uint8 [N,H,W,3], ``labels`` int [N] and optionally
  ``class_names``.

``dataset_info`` / ``get_channel_stats`` / ``get_subset_indices`` / ``create_eval_loader`` / ``create_dataloaders`` keep
the reference's names and argument meaning.  Loader workers, shuffling, ``drop_last`` and pinned memory as :158-166.
"""
from __future__ import annotations

import os
from functools import lru_cache

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .transforms import AugmentTransform, EvalTransform

_NUM_WORKERS = 8
_CHANNEL_STATS_SAMPLES = 5000
_IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".ppm", ".webp")
_EVAL_SPLITS = ("validation", "val", "test")


def is_local_dataset(name) -> bool:
    return isinstance(name, str) and os.path.isdir(name)


def _split_source(root: str, split: str):
    if os.path.isfile(os.path.join(root, split + ".npz")):
        return "npz", os.path.join(root, split + ".npz")
    if os.path.isdir(os.path.join(root, split)):
        return "dir", os.path.join(root, split)
    return None, None


class LocalImageSplit(Dataset):
    """uint8 ``[3, H, W]`` images + labels of one split; ``transform(index, image) -> dict`` builds the sample"""

    def __init__(self, root: str, split: str, class_names=None):
        kind, path = _split_source(root, split)
        if kind is None:
            raise FileNotFoundError(f"{root}: no split {split!r} (expected {split}.npz or a {split}/ directory)")
        self.kind = kind
        if kind == "npz":
            with np.load(path, allow_pickle=False) as z:
                self.images = z["images"]
                self.labels = z["labels"].astype(np.int64)
                names = [str(n) for n in z["class_names"]] if "class_names" in z.files else None
            if self.images.ndim != 4 or self.images.shape[-1] != 3 or self.images.dtype != np.uint8:
                raise ValueError(f"{path}: images must be uint8 [N, H, W, 3], got {self.images.dtype} {self.images.shape}")
            self.class_names = tuple(names if names is not None else
                                     (class_names or [str(i) for i in range(int(self.labels.max()) + 1)]))
        else:
            classes = sorted(d for d in os.listdir(path) if os.path.isdir(os.path.join(path, d)))
            self.class_names = tuple(class_names or classes)
            index = {c: i for i, c in enumerate(self.class_names)}
            self.files, labels = [], []
            for c in classes:
                for f in sorted(os.listdir(os.path.join(path, c))):
                    if f.lower().endswith(_IMAGE_EXT):
                        self.files.append(os.path.join(path, c, f))
                        labels.append(index[c])
            self.labels = np.asarray(labels, dtype=np.int64)
        self.transform = None

    def __len__(self):
        return len(self.labels)

    def set_epoch(self, epoch: int) -> None:
        """forwarded to the per-sample transform (``Trainer.train`` calls it before every epoch)"""
        if hasattr(self.transform, "set_epoch"):
            self.transform.set_epoch(epoch)

    def image(self, i: int) -> torch.Tensor:
        if self.kind == "npz":
            arr = self.images[i]
        else:
            from PIL import Image
            with Image.open(self.files[i]) as im:
                arr = np.asarray(im.convert("RGB"))
        return torch.from_numpy(np.array(arr, copy=True)).permute(2, 0, 1).contiguous()

    def __getitem__(self, i: int):
        img, label = self.image(i), int(self.labels[i])
        return self.transform(i, img, label) if self.transform is not None else {"image": img, "label": label}


@lru_cache(maxsize=None)
def dataset_info(dataset_name: str) -> dict:
    """reference :24-44 for a local root: class names from the training split, evaluation split by the same preference
    order (validation, then test, then train)"""
    train = LocalImageSplit(dataset_name, "train")
    eval_split = next((s for s in _EVAL_SPLITS if _split_source(dataset_name, s)[0] is not None), "train")
    return {"image_key": "image", "label_key": "label", "num_classes": len(train.class_names),
            "class_names": train.class_names, "train_split": "train", "eval_split": eval_split}


@lru_cache(maxsize=None)
def get_channel_stats(dataset_name: str):
    """per-channel mean / std of the first 5 000 training images in [0, 1], pooled over pixels with the parallel-variance
    update of the reference (:47-71)"""
    ds = LocalImageSplit(dataset_name, "train")
    mean, m2, count = np.zeros(3), np.zeros(3), 0
    for i in range(min(len(ds), _CHANNEL_STATS_SAMPLES)):
        flat = ds.image(i).permute(1, 2, 0).reshape(-1, 3).numpy().astype(np.float64) / 255.0
        n = flat.shape[0]
        batch_mean, batch_var = flat.mean(axis=0), flat.var(axis=0)
        delta = batch_mean - mean
        new_count = count + n
        mean = mean + delta * n / new_count
        m2 = m2 + batch_var * n + delta ** 2 * count * n / new_count
        count = new_count
    std = np.sqrt(m2 / count)
    return tuple(mean.tolist()), tuple(std.tolist())


def get_subset_indices(dataset_name: str, parent_name: str):
    """reference :74-80: positions of this dataset's classes in the parent's class list (None if the sets are equal)"""
    child, parent = dataset_info(dataset_name)["class_names"], dataset_info(parent_name)["class_names"]
    if set(child) == set(parent):
        return None
    parent_map = {name: idx for idx, name in enumerate(parent)}
    return tuple(parent_map[name] for name in child)


def build_eval_transform(image_size: int, *, mean, std, crop_ratio: float) -> EvalTransform:
    return EvalTransform(image_size, mean=mean, std=std, crop_ratio=crop_ratio)


def _collate(samples):
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        out[k] = torch.stack(v) if isinstance(v[0], torch.Tensor) else torch.tensor(v, dtype=torch.int64)
    return out


def _loader(ds, batch_size, *, shuffle, drop_last, num_workers, seed):
    g = torch.Generator().manual_seed(seed)
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=drop_last,
                      pin_memory=torch.cuda.is_available(), persistent_workers=num_workers > 0, collate_fn=_collate,
                      generator=g)


def create_eval_loader(dataset_name: str, *, image_size: int, batch_size: int, mean, std, crop_ratio: float,
                       num_workers: int = _NUM_WORKERS, class_names=None) -> DataLoader:
    """reference :97-123: ``{"pixel_values", "label"}`` batches of the evaluation split, not shuffled"""
    info = dataset_info(dataset_name)
    tf = build_eval_transform(image_size, mean=mean, std=std, crop_ratio=crop_ratio)
    ds = LocalImageSplit(dataset_name, info["eval_split"], class_names=class_names or info["class_names"])
    ds.transform = _EvalView(tf)
    return _loader(ds, batch_size, shuffle=False, drop_last=False, num_workers=num_workers, seed=0)


class _EvalView:
    def __init__(self, tf):
        self.tf = tf

    def __call__(self, i: int, img: torch.Tensor, label: int):
        return {"pixel_values": self.tf(img), "label": label}


class _DualView:
    """per-sample transform of the training split; the augmentation RNG of sample i in epoch e is seeded from
    (seed, e, i): reproducible whatever the worker count"""

    def __init__(self, clean_tf, aug_tf, seed: int):
        self.clean_tf, self.aug_tf, self.seed = clean_tf, aug_tf, seed
        # the epoch lives in SHARED memory: DataLoader workers (persistent ones included) hold a pickled copy of this
        # object, and a plain attribute set in the parent would never reach them -- every epoch would then replay the
        # same crop / flip / TrivialAugment op for sample i (the reference draws fresh ones each epoch,
        # src/data/datasets.py:137-149 with torchvision's global RNG)
        self._epoch = torch.zeros(1, dtype=torch.int64).share_memory_()

    @property
    def epoch(self) -> int:
        return int(self._epoch[0])

    def set_epoch(self, epoch: int) -> None:
        self._epoch[0] = int(epoch)

    def __call__(self, i: int, img: torch.Tensor, label: int):
        gen = torch.Generator().manual_seed((self.seed * 1_000_003 + self.epoch) * 2_000_003 + i)
        return {"clean": self.clean_tf(img), "augmented": self.aug_tf(img, gen), "label": label}


def create_dataloaders(config, *, teacher_stats, num_workers: int = _NUM_WORKERS):
    """reference :126-178: (train loader of dual-view batches, evaluation loader).  ``teacher_stats`` = (mean, std) the
    frozen teacher was trained with (``TeacherModel.mean / .std``)."""
    name = config.data.dataset
    info = dataset_info(name)
    mean, std = get_channel_stats(name)
    image_size = config.model.vit.img_size
    crop_ratio = float(config.data.eval_crop_ratio)
    aug_tf = AugmentTransform(image_size, mean=mean, std=std)
    teacher_mean, teacher_std = teacher_stats
    clean_tf = build_eval_transform(image_size, mean=teacher_mean, std=teacher_std, crop_ratio=crop_ratio)
    seed = int(config.run.get("seed", 0)) if hasattr(config, "run") else 0
    train = LocalImageSplit(name, info["train_split"])
    train.transform = _DualView(clean_tf, aug_tf, seed)
    train_loader = _loader(train, config.data.batch_size, shuffle=True, drop_last=True, num_workers=num_workers, seed=seed)
    val_loader = create_eval_loader(name, image_size=image_size, batch_size=config.data.batch_size, mean=mean, std=std,
                                    crop_ratio=crop_ratio, num_workers=num_workers)
    return train_loader, val_loader
