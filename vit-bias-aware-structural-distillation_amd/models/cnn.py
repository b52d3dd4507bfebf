"""Convolutional teachers for the cross-architecture configuration (BASELINE configs[2]: ResNet-50 teacher,
single "layer" of 7 x 7 = 49 tokens x 2048 channels, uniform importance).

The reference takes CNN teachers from timm (``src/models/teacher.py:118``; its cross-arch overlay names a
ConvNeXt, ``configs/experiment/basd_imagenet_cross_arch.yaml:6``) and only ever calls ``forward_features`` on them
(``teacher.py:184-191``).  timm / torchvision are not available offline, so the trunk is defined here with the
torchvision / timm parameter names (``conv1``, ``bn1``, ``layer1..4.N.conv1`` ...) so that a local state dict of
either package loads.  The frozen trunk is a black box for the BASD step ("c3"): it runs through
PyTorch-ROCm / MIOpen in channels-last bf16, no hand-written kernel is involved.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    """ResNet-v1.5 trunk (stride on the 3x3 convolution), stages ``layer1..layer4`` as in torchvision / timm."""

    def __init__(self, depths=(3, 4, 6, 3), num_classes: int = 0, in_chans: int = 3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_chans, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        for i, (planes, depth) in enumerate(zip((64, 128, 256, 512), depths)):
            setattr(self, f"layer{i + 1}", self._stage(planes, depth, stride=1 if i == 0 else 2))
        self.num_features = self.embed_dim = 512 * Bottleneck.expansion
        self.fc = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, depth, stride):
        down = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * Bottleneck.expansion))
        blocks = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * Bottleneck.expansion
        blocks += [Bottleneck(self.inplanes, planes) for _ in range(depth - 1)]
        return nn.Sequential(*blocks)

    def forward_features(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))        # [B, 2048, H/32, W/32]

    def forward(self, x):
        return self.fc(self.forward_features(x).mean(dim=(2, 3)))


CNN_PRESETS = {
    "resnet50": lambda: ResNet((3, 4, 6, 3)),
    "resnet101": lambda: ResNet((3, 4, 23, 3)),
}


def create_cnn(name: str) -> nn.Module:
    if name not in CNN_PRESETS:
        raise ValueError(f"unknown CNN preset {name!r}; known: {sorted(CNN_PRESETS)}")
    return CNN_PRESETS[name]()
