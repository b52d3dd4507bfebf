"""Loader for the reference's Hydra/OmegaConf config tree without hydra.

Accepts ``configs/config.yaml`` of the reference unchanged (keys: ``run``,
``data``, ``model``, ``checkpoint``, ``training``, ``basd``; reference
configs/config.yaml:1-44), an optional ``experiment`` overlay
(configs/experiment/*.yaml) and ``a.b=c`` overrides, and evaluates the three
resolvers of ``src/resolvers.py:6-21`` (``num_classes``, ``label_smoothing``,
``eval_crop_ratio``) plus plain ``${a.b}`` references.  hydra-core / omegaconf
are not installed in the build image.
"""
from __future__ import annotations

import os
import re

import yaml

# reference src/data/datasets.py:24-43 needs the HF builder (network); the class counts of
# the datasets its configs name are fixed facts
_NUM_CLASSES = {
    "ILSVRC/imagenet-1k": 1000, "imagenet-1k": 1000, "uoft-cs/cifar100": 100, "cifar100": 100,
    "uoft-cs/cifar10": 10, "synthetic": 1000,
}


class Config(dict):
    """dict with attribute access (enough of DictConfig for the train step)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(base: dict, over: dict) -> dict:
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def _lookup(root: dict, path: str):
    cur = root
    for part in path.split("."):
        cur = cur[part]
    return cur


def _num_classes(name: str) -> int:
    import os
    if os.path.isdir(name):                 # a local dataset root: the class folders / class_names of its training split
        from .data import dataset_info      # (reference src/resolvers.py:10 asks the dataset builder the same question)
        return dataset_info(name)["num_classes"]
    if name not in _NUM_CLASSES:
        raise KeyError(f"dataset {name!r}: number of classes unknown offline; set model.num_classes explicitly")
    return _NUM_CLASSES[name]


_RESOLVERS = {
    "num_classes": lambda ds: _num_classes(str(ds)),
    "label_smoothing": lambda ds: 1.0 / _num_classes(str(ds)),
    "eval_crop_ratio": lambda img, patch: float(img) / (float(img) + 2 * float(patch)),
}
_INNER = re.compile(r"\$\{([^${}]+)\}")
_FLOAT = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def _resolve_str(s: str, root: dict):
    while True:
        m = _INNER.search(s)
        if not m:
            return s
        expr = m.group(1)
        if ":" in expr:
            name, args = expr.split(":", 1)
            val = _RESOLVERS[name.strip()](*[a.strip() for a in args.split(",")])
        else:
            val = _resolve(_lookup(root, expr.strip()), root)
        if m.start() == 0 and m.end() == len(s):
            return val
        s = s[:m.start()] + str(val) + s[m.end():]


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str) and "${" in node:
        return _resolve_str(node, root)
    if isinstance(node, str) and _FLOAT.match(node):     # YAML 1.1 reads "5e-4" as a string
        return float(node)
    return node


def load_config(path: str, experiment: str | None = None, overrides: list[str] | None = None) -> Config:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg.pop("defaults", None)
    cfg.pop("hydra", None)
    if experiment:
        exp_path = experiment if os.path.exists(experiment) else os.path.join(
            os.path.dirname(path), "experiment", f"{experiment}.yaml")
        with open(exp_path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    for ov in overrides or []:
        key, val = ov.split("=", 1)
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return _wrap(_resolve(cfg, cfg))
