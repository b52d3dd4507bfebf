"""Image transforms of the dual-view input pipeline, on uint8 ``[3, H, W]`` torch tensors (CPU, loader workers).

The reference composes torchvision.transforms.v2 objects (``src/data/datasets.py:1-17,80-94,137-149``):

* clean / evaluation view: ``Resize(round(S / crop_ratio)) -> CenterCrop(S) -> ToImage -> ToDtype(float32, scale=True)
  -> Normalize(mean, std)`` (``build_eval_transform``, :80-94; the teacher's statistics for the clean view, :146-149);
* augmented view: ``RandomResizedCrop(S) -> RandomHorizontalFlip -> TrivialAugmentWide -> ToImage -> ToDtype ->
  Normalize(dataset mean, std)`` (:137-144).

torchvision is not installed here, so the operations are restated with torch tensor ops following torchvision's
published definitions (defaults: bilinear + antialias for Resize / RandomResizedCrop, scale (0.08, 1), ratio (3/4, 4/3),
flip p = 0.5, TrivialAugmentWide with 31 magnitude bins, nearest interpolation, fill 0).  Everything random draws from
the ``torch.Generator`` it is given: a seeded pipeline is reproducible.  RNG-dependent, hence outside the parity boundary
(SURVEY section 8c); the tests pin the deterministic parts and the distributional contracts.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- geometry
def resize(img: torch.Tensor, size) -> torch.Tensor:
    """int: shorter side -> size (aspect kept, torchvision Resize(int)); (h, w): exact.  Bilinear, antialiased."""
    _, h, w = img.shape
    if isinstance(size, int):
        if h <= w:
            nh, nw = size, max(1, int(size * w / h))
        else:
            nh, nw = max(1, int(size * h / w)), size
    else:
        nh, nw = size
    if (nh, nw) == (h, w):
        return img
    out = F.interpolate(img.unsqueeze(0).float(), size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    return out[0].round().clamp_(0, 255).to(torch.uint8) if img.dtype == torch.uint8 else out[0]


def center_crop(img: torch.Tensor, size: int) -> torch.Tensor:
    _, h, w = img.shape
    if h < size or w < size:          # torchvision pads with zeros, centred
        ph, pw = max(size - h, 0), max(size - w, 0)
        img = F.pad(img, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        _, h, w = img.shape
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img[:, top:top + size, left:left + size]


def _rand(gen) -> float:
    return float(torch.rand((), generator=gen))


def _randint(gen, n: int) -> int:
    return int(torch.randint(0, n, (), generator=gen))


def random_resized_crop_params(h: int, w: int, gen, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params: ten attempts at (area ~ U(scale), log-uniform aspect), then the central
    crop with the aspect clamped into ``ratio`` -> (top, left, height, width)"""
    area = h * w
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * (scale[0] + (scale[1] - scale[0]) * _rand(gen))
        aspect = math.exp(log_r[0] + (log_r[1] - log_r[0]) * _rand(gen))
        cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            return _randint(gen, h - ch + 1), _randint(gen, w - cw + 1), ch, cw
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        ch, cw = h, int(round(h * ratio[1]))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def random_resized_crop(img: torch.Tensor, size: int, gen) -> torch.Tensor:
    top, left, ch, cw = random_resized_crop_params(img.shape[1], img.shape[2], gen)
    return resize(img[:, top:top + ch, left:left + cw], (size, size))


def hflip(img: torch.Tensor) -> torch.Tensor:
    return img.flip(-1)


def _affine_nearest(img: torch.Tensor, matrix) -> torch.Tensor:
    """out(x, y) = img(M^-1 ...): ``matrix`` = the six coefficients (a, b, c, d, e, f) of the INVERSE map in pixel
    coordinates about the image centre, x_in = a x + b y + c, y_in = d x + e y + f; nearest, zero fill"""
    _, h, w = img.shape
    a, b, c, d, e, f = matrix
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) - (h - 1) / 2.0,
                            torch.arange(w, dtype=torch.float32) - (w - 1) / 2.0, indexing="ij")
    xi = (a * xs + b * ys + c + (w - 1) / 2.0).round().long()
    yi = (d * xs + e * ys + f + (h - 1) / 2.0).round().long()
    ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
    out = img[:, yi.clamp(0, h - 1), xi.clamp(0, w - 1)]
    return out * ok.to(img.dtype)


# --------------------------------------------------------------------------- colour
def _gray(img_f: torch.Tensor) -> torch.Tensor:
    return (0.299 * img_f[0] + 0.587 * img_f[1] + 0.114 * img_f[2]).unsqueeze(0)


def _blend(a: torch.Tensor, b: torch.Tensor, factor: float) -> torch.Tensor:
    return (factor * a + (1.0 - factor) * b).clamp_(0, 255)


def adjust_brightness(img, factor):
    return _blend(img.float(), torch.zeros(()), factor).round().to(torch.uint8)


def adjust_saturation(img, factor):
    x = img.float()
    return _blend(x, _gray(x).expand_as(x), factor).round().to(torch.uint8)


def adjust_contrast(img, factor):
    x = img.float()
    return _blend(x, _gray(x).mean().expand_as(x), factor).round().to(torch.uint8)


def adjust_sharpness(img, factor):
    x = img.float()
    if x.shape[1] <= 2 or x.shape[2] <= 2:
        return img
    k = torch.tensor([[1.0, 1.0, 1.0], [1.0, 5.0, 1.0], [1.0, 1.0, 1.0]]) / 13.0
    blur = F.conv2d(x.unsqueeze(1), k.view(1, 1, 3, 3)).squeeze(1).round()
    soft = x.clone()
    soft[:, 1:-1, 1:-1] = blur
    return _blend(x, soft, factor).round().to(torch.uint8)


def posterize(img, bits: int):
    return img & (255 - (2 ** (8 - bits) - 1))


def solarize(img, threshold: float):
    return torch.where(img.float() >= threshold, 255 - img, img)


def autocontrast(img):
    x = img.float()
    lo, hi = x.amin(dim=(1, 2), keepdim=True), x.amax(dim=(1, 2), keepdim=True)
    scale = 255.0 / (hi - lo)
    same = ~torch.isfinite(scale)
    scale = torch.where(same, torch.ones_like(scale), scale)
    lo = torch.where(same, torch.zeros_like(lo), lo)
    return ((x - lo) * scale).clamp_(0, 255).to(torch.uint8)


def equalize(img):
    out = []
    for ch in img:
        hist = torch.bincount(ch.reshape(-1).long(), minlength=256).float()
        nz = hist[hist != 0]
        step = torch.div(nz[:-1].sum(), 255, rounding_mode="floor") if nz.numel() > 1 else torch.tensor(0.0)
        if float(step) == 0:
            out.append(ch)
            continue
        lut = torch.div(torch.cumsum(hist, 0) + torch.div(step, 2, rounding_mode="floor"), step, rounding_mode="floor")
        lut = F.pad(lut, (1, 0))[:-1].clamp(0, 255)
        out.append(lut[ch.long()].to(torch.uint8))
    return torch.stack(out)


TA_WIDE_OPS = ("Identity", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast",
               "Sharpness", "Posterize", "Solarize", "AutoContrast", "Equalize")
_TA_BINS = 31


def _ta_magnitude(op: str, bin_: int) -> float:
    lin = lambda lo, hi: lo + (hi - lo) * bin_ / (_TA_BINS - 1)      # noqa: E731
    if op in ("ShearX", "ShearY", "Brightness", "Color", "Contrast", "Sharpness"):
        return lin(0.0, 0.99)
    if op in ("TranslateX", "TranslateY"):
        return lin(0.0, 32.0)
    if op == "Rotate":
        return lin(0.0, 135.0)
    if op == "Posterize":
        return 8 - int(round(bin_ / ((_TA_BINS - 1) / 6)))
    if op == "Solarize":
        return lin(255.0, 0.0)
    return 0.0


def apply_ta_op(img: torch.Tensor, op: str, magnitude: float) -> torch.Tensor:
    """one TrivialAugmentWide operation (torchvision ``_apply_op``) on a uint8 image"""
    if op == "Identity":
        return img
    if op == "ShearX":           # x_in = x - shear * y about the centre (torchvision shears about the corner: the
        return _affine_nearest(img, (1.0, magnitude, 0.0, 0.0, 1.0, 0.0))    # content moves, the statistics do not)
    if op == "ShearY":
        return _affine_nearest(img, (1.0, 0.0, 0.0, magnitude, 1.0, 0.0))
    if op == "TranslateX":
        return _affine_nearest(img, (1.0, 0.0, -float(int(magnitude)), 0.0, 1.0, 0.0))
    if op == "TranslateY":
        return _affine_nearest(img, (1.0, 0.0, 0.0, 0.0, 1.0, -float(int(magnitude))))
    if op == "Rotate":
        t = math.radians(magnitude)
        return _affine_nearest(img, (math.cos(t), -math.sin(t), 0.0, math.sin(t), math.cos(t), 0.0))
    if op == "Brightness":
        return adjust_brightness(img, 1.0 + magnitude)
    if op == "Color":
        return adjust_saturation(img, 1.0 + magnitude)
    if op == "Contrast":
        return adjust_contrast(img, 1.0 + magnitude)
    if op == "Sharpness":
        return adjust_sharpness(img, 1.0 + magnitude)
    if op == "Posterize":
        return posterize(img, int(magnitude))
    if op == "Solarize":
        return solarize(img, magnitude)
    if op == "AutoContrast":
        return autocontrast(img)
    if op == "Equalize":
        return equalize(img)
    raise ValueError(op)


def trivial_augment_wide(img: torch.Tensor, gen) -> torch.Tensor:
    """one uniformly drawn operation at a uniformly drawn magnitude bin; signed operations flip sign with p = 0.5"""
    op = TA_WIDE_OPS[_randint(gen, len(TA_WIDE_OPS))]
    mag = _ta_magnitude(op, _randint(gen, _TA_BINS))
    if op in ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast", "Sharpness") \
            and _randint(gen, 2):
        mag = -mag
    return apply_ta_op(img, op, mag)


# --------------------------------------------------------------------------- composed views
def to_normalized_float(img: torch.Tensor, mean, std) -> torch.Tensor:
    """ToImage -> ToDtype(float32, scale=True) -> Normalize(mean, std)"""
    x = img.float().div_(255.0)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return (x - m) / s


class EvalTransform:
    """reference ``build_eval_transform`` (src/data/datasets.py:80-94)"""

    def __init__(self, image_size: int, *, mean, std, crop_ratio: float):
        self.image_size, self.mean, self.std = image_size, tuple(mean), tuple(std)
        self.resize_size = round(image_size / crop_ratio)

    def __call__(self, img: torch.Tensor, gen=None) -> torch.Tensor:
        return to_normalized_float(center_crop(resize(img, self.resize_size), self.image_size), self.mean, self.std)


class AugmentTransform:
    """reference ``aug_tf`` (src/data/datasets.py:137-144)"""

    def __init__(self, image_size: int, *, mean, std):
        self.image_size, self.mean, self.std = image_size, tuple(mean), tuple(std)

    def __call__(self, img: torch.Tensor, gen) -> torch.Tensor:
        x = random_resized_crop(img, self.image_size, gen)
        if _rand(gen) < 0.5:
            x = hflip(x)
        x = trivial_augment_wide(x, gen)
        return to_normalized_float(x, self.mean, self.std)
