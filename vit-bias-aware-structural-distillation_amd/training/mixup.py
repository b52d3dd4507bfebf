"""On-device RandomChoice([MixUp(alpha=1), CutMix(alpha=1)]) producing soft targets
(reference src/training/trainer.py:89-92,138 uses torchvision.transforms.v2, which is not
installed here).  RNG dependent, hence outside the parity boundary."""
from __future__ import annotations

import torch


def mixup_cutmix(images: torch.Tensor, targets: torch.Tensor, num_classes: int, generator=None,
                 out: torch.Tensor | None = None, out_targets: torch.Tensor | None = None):
    """Partner of sample i is sample i - 1 (a roll by one, as torchvision's v2 transforms do).  The rolled batch is never
    materialised: the blend / the pasted box reads the two shifted slices of ``images`` directly and writes ``out``
    (e.g. the static input buffer of the captured step) in one pass.  Returns (mixed images, soft targets)."""
    if out is None:
        out = torch.empty_like(images)
    lam = float(torch.distributions.Beta(1.0, 1.0).sample())
    if float(torch.rand((), generator=generator)) < 0.5:          # MixUp: lam * x_i + (1 - lam) * x_{i-1}
        torch.lerp(images[:-1], images[1:], lam, out=out[1:])
        torch.lerp(images[-1:], images[:1], lam, out=out[:1])
    else:                                                          # CutMix
        h, w = images.shape[-2:]
        r = (1.0 - lam) ** 0.5
        ch, cw = int(h * r), int(w * r)
        cy = int(torch.randint(h, (1,), generator=generator))
        cx = int(torch.randint(w, (1,), generator=generator))
        y0, y1 = max(cy - ch // 2, 0), min(cy + ch // 2, h)
        x0, x1 = max(cx - cw // 2, 0), min(cx + cw // 2, w)
        out.copy_(images)
        out[1:, ..., y0:y1, x0:x1] = images[:-1, ..., y0:y1, x0:x1]
        out[:1, ..., y0:y1, x0:x1] = images[-1:, ..., y0:y1, x0:x1]
        lam = 1.0 - (y1 - y0) * (x1 - x0) / float(h * w)
    onehot = torch.nn.functional.one_hot(targets, num_classes).float()
    if out_targets is None:
        out_targets = torch.empty_like(onehot)
    torch.lerp(onehot.roll(1, 0), onehot, lam, out=out_targets)
    return out, out_targets
