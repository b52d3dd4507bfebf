"""Teacher handling: architecture probe, per-block token + importance tap.

Mirrors the operator surface of reference ``src/models/teacher.py``
(``TeacherModel`` :9-20, ``probe_model`` :42-110, ``load_teacher`` :113-148,
``_to_token_format`` :151-158, ``estimate_intrinsic_dim`` :161-177,
``extract_intermediates`` :180-216).  The reference's full-map attention hook
(``make_attn_capture_hook`` :27-39) has no counterpart in the product: the blocks emit
the CLS-row importance themselves; a reference-style hook lives in ``tests/`` as a checker.

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


_STAGE_CONTAINERS = ("blocks", "layers", "stages")


def _stage_paths(model: nn.Module) -> list[str]:
    """Module paths of the model's repeated stages, in execution order: the children of a ``blocks`` / ``layers`` /
    ``stages`` container (ViT, Swin, ConvNeXt: what the reference's probe looks for, teacher.py:46-50), or the numbered
    attribute family ``layer1, layer2, ...`` of torchvision / timm ResNets, which the reference's probe cannot see
    ("c3")."""
    children = dict(model.named_children())
    for name in _STAGE_CONTAINERS:
        box = children.get(name)
        if isinstance(box, (nn.Sequential, nn.ModuleList)) and len(box) > 0:
            return [f"{name}.{i}" for i in range(len(box))]
    numbered = sorted((int(n[5:]), n) for n in children if n.startswith("layer") and n[5:].isdigit())
    if numbered and [i for i, _ in numbered] == list(range(numbered[0][0], numbered[0][0] + len(numbered))):
        return [n for _, n in numbered]
    raise ValueError("probe_model: no block container (blocks / layers / stages) and no layer1..N stage family found")


def probe_model(model: nn.Module, img_size: int) -> dict:
    """Architecture record with the reference's keys (teacher.py:100-110): ``embed_dim, heads_per_layer, depth,
    mlp_ratio, layer_paths, attn_subpath, has_cls_token, feature_format, num_tokens``.  One pass over the stage
    modules collects heads / MLP width, one forward of a zero image through a hook on the LAST stage tells the
    feature format (token / nchw / nhwc) and the token count.  Runs on the model's own device."""
    layer_paths = _stage_paths(model)
    width = getattr(model, "embed_dim", None) or getattr(model, "num_features", None)
    heads, attn_name, hidden = [], None, None
    for path in layer_paths:
        stage = model.get_submodule(path)
        attn = next(((n, m) for n, m in stage.named_children() if hasattr(m, "num_heads")), None)
        heads.append(attn[1].num_heads if attn else 0)
        attn_name = attn_name or (attn[0] if attn else None)
        if hidden is None:
            hidden = next((m.fc1.out_features for m in stage.children() if hasattr(m, "fc1")), None)
    has_cls = isinstance(getattr(model, "cls_token", None), nn.Parameter)
    seen = []
    hook = model.get_submodule(layer_paths[-1]).register_forward_hook(lambda m, i, o: seen.append(o))
    first = next(model.parameters())
    mode = model.training
    model.eval()
    try:
        with torch.no_grad():
            model(torch.zeros(1, 3, img_size, img_size, device=first.device, dtype=first.dtype))
    finally:
        hook.remove()
        model.train(mode)
    out = seen[-1]
    if out.dim() == 4:                      # conv feature map: channels are the long axis next to the batch (nchw) or last
        fmt, n_tok = ("nchw" if out.shape[1] > out.shape[3] else "nhwc"), 0
        heads = [1]
        width = width or (out.shape[1] if fmt == "nchw" else out.shape[3])
    else:
        fmt, n_tok = "token", out.shape[1] - int(has_cls)
    return {
        "embed_dim": width, "heads_per_layer": heads, "depth": len(layer_paths),
        "mlp_ratio": (hidden / width) if hidden else 0.0, "layer_paths": layer_paths, "attn_subpath": attn_name,
        "has_cls_token": has_cls, "feature_format": fmt, "num_tokens": n_tok,
    }


@torch.no_grad()
def fold_layerscale(model: nn.Module) -> int:
    """Frozen LayerScale teachers (DINOv2, the reference's default ``dinov2_vitb14``, configs/config.yaml:38):
    ``gamma * (W x + b)`` is the linear layer ``(diag(gamma) W) x + gamma b``, so every ``ls1`` / ``ls2`` is folded
    into the projection in front of it (exact in fp32, done before the bf16 cast) and replaced by ``nn.Identity``:
    the blocks then take the fused inference path (residual add + LayerNorm kernel, GEMM epilogues) like a plain ViT.
    Returns the number of folded scales."""
    folded = 0
    for block in model.modules():
        for ls_name, lin_path in (("ls1", "attn.proj"), ("ls2", "mlp.fc2")):
            ls = getattr(block, ls_name, None)
            if ls is None or isinstance(ls, nn.Identity) or not hasattr(ls, "gamma"):
                continue
            try:
                lin = block.get_submodule(lin_path)
            except AttributeError:
                continue
            if not isinstance(lin, nn.Linear) or lin.weight.requires_grad:
                continue
            gamma = ls.gamma.detach().to(lin.weight.dtype)
            lin.weight.mul_(gamma.unsqueeze(1))
            if lin.bias is not None:
                lin.bias.mul_(gamma)
            setattr(block, ls_name, nn.Identity())
            folded += 1
    return folded


def load_teacher(model_name: str, img_size: int, *, weights: str | None = None, device="cuda",
                 seed: int = 42, patch_size: int | None = None, dtype=torch.bfloat16) -> TeacherModel:
    """Build (not download: reference teacher.py:113-148 fetches from the network) the named teacher -- a ViT preset of
    ``models/vit.py`` or a CNN preset of ``models/cnn.py`` -- frozen, in eval mode, weights pre-cast to bf16; ``weights``
    = a local state dict (timm / torchvision parameter names)."""
    from .cnn import CNN_PRESETS, create_cnn
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    if model_name in VIT_PRESETS:
        model = create_vit(model_name, num_classes=0, img_size=img_size, patch_size=patch_size)
    elif model_name in CNN_PRESETS:
        model = create_cnn(model_name)
    else:
        raise ValueError(f"teacher {model_name!r}: not a known preset ({sorted(VIT_PRESETS) + sorted(CNN_PRESETS)}); "
                         "wrap your own module in TeacherModel(...) with the fields probe_model() returns")
    torch.random.set_rng_state(gen_state)
    if weights:
        state = torch.load(weights, map_location="cpu", weights_only=True)
        model.load_state_dict(state.get("model_state_dict", state), strict=False)
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    fold_layerscale(model)
    info = probe_model(model, img_size)
    model = model.to(dtype)
    if info["feature_format"] != "token" and torch.device(device).type == "cuda":
        model = model.to(memory_format=torch.channels_last)
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
    """Marchenko-Pastur rank of the teacher's LAST-stage tokens over a calibration batch (reference
    teacher.py:161-177; feeds ``_derive_from_teacher``, src/train.py:57-66).  The last stage's tokens are exactly what
    ``extract_intermediates`` returns for its highest layer index, so the tap machinery is reused instead of a second
    hook; the rank is computed on the GPU (token Gram in fp64 -> blocked or LDS-resident eigensolver -> device count)."""
    tokens, _ = extract_intermediates(teacher, images)
    last = tokens[max(tokens)]
    return marchenko_pastur_rank(last.reshape(-1, last.shape[-1]).float())


@torch.no_grad()
def extract_intermediates(teacher: TeacherModel, x: torch.Tensor, on_layer=None):
    """-> (tokens {j: [B,N,D]}, importance {j: [B,N]}); CNN teachers: one layer, uniform importance.
    ``on_layer(j, tokens_j)`` is called from the block's forward hook, i.e. as soon as layer j's tokens have been
    enqueued (the trainer launches that layer's selector statistics on another stream from it)."""
    x = x.to(next(teacher.model.parameters()).dtype)
    if teacher.feature_format != "token":
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
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
