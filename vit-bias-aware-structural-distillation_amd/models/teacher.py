"""Teacher handling: architecture probe, per-block token + importance tap.

Mirrors the operator surface of reference ``src/models/teacher.py``
(``TeacherModel`` :9-20, ``make_attn_capture_hook`` :27-39, ``probe_model``
:42-110, ``load_teacher`` :113-148, ``_to_token_format`` :151-158,
``estimate_intrinsic_dim`` :161-177, ``extract_intermediates`` :180-216).

* ``extract_intermediates`` returns ``(tokens {j: [B,N,D]}, importance
  {j: [B,N]})``: the second dict holds the head-averaged CLS-row (or
  query-mean) importance each block computes from its own q/k, which is all the
  loss ever reads of the reference's ``[B,H,T,T]`` maps (5.7 GB at the
  benchmark config, never materialised here).  ``BASDLoss`` accepts either.
* ``load_teacher`` cannot fetch weights (no network): it builds the named
  architecture with seeded random weights, or loads a local state dict when
  ``weights`` is given.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from ..losses.layer_selector import marchenko_pastur_rank
from .vit import VIT_PRESETS, create_vit


class TeacherModel(NamedTuple):
    model: torch.nn.Module
    embed_dim: int
    heads_per_layer: list
    depth: int
    mlp_ratio: float
    layer_paths: list
    attn_subpath: str | None
    has_cls_token: bool
    feature_format: str
    mean: tuple
    std: tuple


_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def make_attn_capture_hook(capture_dict: dict, layer_idx: int, *, apply_softmax: bool = True):
    """Reference-compatible forward hook that stores the FULL map (teacher.py:27-39).

    Kept for callers of the reference surface / for tests; the train step uses the
    in-block importance tap instead.
    """
    def hook(mod, inp, out):
        x_in = inp[0]
        b, n, c = x_in.shape
        nh = mod.num_heads
        hd = c // nh
        qkv = mod.qkv(x_in).reshape(b, n, 3, nh, hd).permute(2, 0, 3, 1, 4)
        attn = (qkv[0] @ qkv[1].transpose(-2, -1)) * (hd ** -0.5)
        capture_dict[layer_idx] = attn.softmax(dim=-1) if apply_softmax else attn
    return hook


def probe_model(model: nn.Module, img_size: int) -> dict:
    """Same keys as the reference probe (teacher.py:100-110); runs on the model's own device."""
    embed_dim = getattr(model, "embed_dim", None) or getattr(model, "num_features", None)
    layer_paths = []
    for name in ("blocks", "layers", "stages"):
        container = getattr(model, name, None)
        if isinstance(container, (nn.Sequential, nn.ModuleList)):
            layer_paths = [f"{name}.{i}" for i in range(len(container))]
            break
    attn_subpath, heads_per_layer, mlp_ratio = None, [], 0.0
    for path in layer_paths:
        block = model.get_submodule(path)
        block_heads = 0
        for child_name, child in block.named_children():
            if hasattr(child, "num_heads"):
                attn_subpath = attn_subpath or child_name
                block_heads = child.num_heads
                break
        heads_per_layer.append(block_heads)
        if mlp_ratio == 0.0:
            for _, child in block.named_children():
                if hasattr(child, "fc1"):
                    mlp_ratio = child.fc1.out_features / embed_dim
                    break
    has_cls_token = any(n == "cls_token" for n, _ in model.named_parameters())
    dev = next(model.parameters()).device
    probe = torch.zeros(1, 3, img_size, img_size, device=dev)
    num_tokens = 0
    was_training = model.training
    model.eval()
    with torch.no_grad():
        captured = {}
        mod = model.get_submodule(layer_paths[-1])
        h = mod.register_forward_hook(lambda m, i, o: captured.update(out=o))
        model(probe)
        h.remove()
        out = captured["out"]
        if out.dim() == 4:
            feature_format = "nchw" if out.shape[1] > out.shape[3] else "nhwc"
        else:
            feature_format = "token"
            num_tokens = out.shape[1] - int(has_cls_token)
    model.train(was_training)
    if feature_format != "token":
        heads_per_layer = [1]
    return {
        "embed_dim": embed_dim, "heads_per_layer": heads_per_layer, "depth": len(layer_paths),
        "mlp_ratio": mlp_ratio, "layer_paths": layer_paths, "attn_subpath": attn_subpath,
        "has_cls_token": has_cls_token, "feature_format": feature_format, "num_tokens": num_tokens,
    }


def load_teacher(model_name: str, img_size: int, *, weights: str | None = None, device="cuda",
                 seed: int = 42, patch_size: int | None = None, dtype=torch.bfloat16) -> TeacherModel:
    """Build (not download) the named teacher, frozen, in eval mode, weights pre-cast to bf16."""
    if model_name not in VIT_PRESETS:
        raise ValueError(f"teacher {model_name!r} is not a known ViT preset (CNN teachers: pass your own "
                         "module to TeacherModel, feature_format 'nchw'/'nhwc')")
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = create_vit(model_name, num_classes=0, img_size=img_size, patch_size=patch_size)
    torch.random.set_rng_state(gen_state)
    if weights:
        state = torch.load(weights, map_location="cpu", weights_only=True)
        model.load_state_dict(state.get("model_state_dict", state), strict=False)
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    info = probe_model(model, img_size)
    model = model.to(dtype)
    for m in model.modules():           # LayerNorm parameters stay fp32 (autocast semantics); the fused
        if isinstance(m, nn.LayerNorm):   # kernel takes bf16 activations with fp32 gamma / beta
            m.float()
    return TeacherModel(model=model, embed_dim=info["embed_dim"], heads_per_layer=info["heads_per_layer"],
                        depth=info["depth"], mlp_ratio=info["mlp_ratio"], layer_paths=info["layer_paths"],
                        attn_subpath=info["attn_subpath"], has_cls_token=info["has_cls_token"],
                        feature_format=info["feature_format"], mean=_IMAGENET_MEAN, std=_IMAGENET_STD)


def _to_token_format(t: torch.Tensor, feature_format: str, has_cls_token: bool) -> torch.Tensor:
    if feature_format == "nhwc":
        t = t.flatten(1, 2)
    elif feature_format == "nchw":
        t = t.flatten(2).transpose(1, 2)
    if has_cls_token:
        t = t[:, 1:, :]
    return t


@torch.no_grad()
def estimate_intrinsic_dim(teacher: TeacherModel, images: torch.Tensor) -> int:
    captured = {}
    mod = teacher.model.get_submodule(teacher.layer_paths[-1])
    h = mod.register_forward_hook(lambda m, i, o: captured.update(out=o))
    teacher.model(images.to(next(teacher.model.parameters()).dtype))
    h.remove()
    tokens = _to_token_format(captured["out"], teacher.feature_format, teacher.has_cls_token)
    return marchenko_pastur_rank(tokens.reshape(-1, tokens.shape[-1]).float())


@torch.no_grad()
def extract_intermediates(teacher: TeacherModel, x: torch.Tensor, on_layer=None):
    """-> (tokens {j: [B,N,D]}, importance {j: [B,N]}); CNN teachers: one layer, uniform importance.
    ``on_layer(j, tokens_j)`` is called from the block's forward hook, i.e. as soon as layer j's tokens have been
    enqueued (the trainer launches that layer's selector statistics on another stream from it)."""
    x = x.to(next(teacher.model.parameters()).dtype)
    if teacher.feature_format != "token":
        feats = _to_token_format(teacher.model.forward_features(x), teacher.feature_format, teacher.has_cls_token)
        b, n, _ = feats.shape
        return {0: feats.contiguous()}, {0: torch.full((b, n), 1.0 / n, device=feats.device)}

    hooks, tokens, taps = [], {}, {}
    for idx, path in enumerate(teacher.layer_paths):
        module = teacher.model.get_submodule(path)

        def make_token_hook(i):
            def hook(mod, inp, out):
                # a view of the block output (CLS stripped by offset): the kernels take batch-strided views
                tokens[i] = _to_token_format(out, teacher.feature_format, teacher.has_cls_token)
                if on_layer is not None:
                    on_layer(i, tokens[i])
            return hook
        hooks.append(module.register_forward_hook(make_token_hook(idx)))
        if teacher.attn_subpath is not None:
            attn_mod = teacher.model.get_submodule(f"{path}.{teacher.attn_subpath}")
            taps[idx] = attn_mod.tap = {"has_cls": teacher.has_cls_token, "out": None}
    try:
        teacher.model(x)
    finally:
        for h in hooks:
            h.remove()
        for idx, path in enumerate(teacher.layer_paths):
            if teacher.attn_subpath is not None:
                teacher.model.get_submodule(f"{path}.{teacher.attn_subpath}").tap = None
    return tokens, {i: t["out"] for i, t in taps.items()}
