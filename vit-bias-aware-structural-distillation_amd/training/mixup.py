"""On-device RandomChoice([MixUp(alpha=1), CutMix(alpha=1)]) producing soft targets
(reference src/training/trainer.py:89-92,138 uses torchvision.transforms.v2, which is not
installed here).  RNG dependent, hence outside the parity boundary."""
from __future__ import annotations

import torch


def mixup_cutmix(images: torch.Tensor, targets: torch.Tensor, num_classes: int, generator=None):
    b = images.shape[0]
    onehot = torch.nn.functional.one_hot(targets, num_classes).float()
    lam = float(torch.distributions.Beta(1.0, 1.0).sample())
    perm_img = images.roll(1, 0)
    perm_tgt = onehot.roll(1, 0)
    if float(torch.rand((), generator=generator)) < 0.5:          # MixUp
        mixed = images * lam + perm_img * (1.0 - lam)
    else:                                                          # CutMix
        h, w = images.shape[-2:]
        r = (1.0 - lam) ** 0.5
        ch, cw = int(h * r), int(w * r)
        cy = int(torch.randint(h, (1,), generator=generator))
        cx = int(torch.randint(w, (1,), generator=generator))
        y0, y1 = max(cy - ch // 2, 0), min(cy + ch // 2, h)
        x0, x1 = max(cx - cw // 2, 0), min(cx + cw // 2, w)
        mixed = images.clone()
        mixed[..., y0:y1, x0:x1] = perm_img[..., y0:y1, x0:x1]
        lam = 1.0 - (y1 - y0) * (x1 - x0) / float(h * w)
    return mixed, onehot * lam + perm_tgt * (1.0 - lam)
