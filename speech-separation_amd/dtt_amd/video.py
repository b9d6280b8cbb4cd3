"""Lip-ROI video encoders.  On the AV forward/backward path but outside the
hand-written-kernel scope of this round:
stock PyTorch-ROCm modules (MIOpen convolutions), same parameter names as the
reference, and NO network fetch (the reference downloads a checkpoint with gdown
in __init__, resnet_video_encoder.py:46-60).

Reference: src/model/VideoEncoders/{resnet,resnet_video_encoder}.py.
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn


_NATIVE_BN = os.environ.get("DTT_VIDEO_NATIVE_BN", "1") == "1"


class _NativeBN:
    """BatchNorm through ATen's own channels-last kernels: MIOpen's NHWC batch-norm kernels took 25 ms of the
    107 ms cfg4 step (10x the HBM time of the tensors)."""

    def forward(self, x):
        if not _NATIVE_BN:
            return super().forward(x)
        with torch.backends.cudnn.flags(enabled=False):
            return super().forward(x)


class BatchNorm2d(_NativeBN, nn.BatchNorm2d):
    pass


class BatchNorm3d(_NativeBN, nn.BatchNorm3d):
    pass


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class BasicBlock(nn.Module):
    """resnet.py:36-81 with relu_type='swish'."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu1 = Swish()
        self.relu2 = Swish()
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = self.relu1(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu2(y + r)


class ResNet18Trunk(nn.Module):
    """resnet.py:84-137, layers [2,2,2,2]."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.layer1 = self._make(64, 2, 1)
        self.layer2 = self._make(128, 2, 2)
        self.layer3 = self._make(256, 2, 2)
        self.layer4 = self._make(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):  # (BatchNorm2d above is a subclass)
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False), BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x).flatten(1)


class VideoEncoder(nn.Module):
    """resnet_video_encoder.py:22-72: Conv3d(5,7,7)+BN3d+Swish+MaxPool3d -> ResNet-18.
    video [B, N, T, H, W] -> [B, N, 512, T]."""

    def __init__(self, checkpoint_path=None):
        super().__init__()
        self.frontend_nout, self.backend_out = 64, 512
        self.trunk = ResNet18Trunk()
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            BatchNorm3d(64), Swish(),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        if checkpoint_path is not None and os.path.exists(checkpoint_path):
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            self.load_state_dict(ck["model_state_dict"], strict=False)

    def forward(self, x):
        B, N, T, H, W = x.shape
        x = self.frontend3D(x.reshape(B * N, 1, T, H, W))
        Tn = x.shape[2]
        x = x.transpose(1, 2).reshape(B * N * Tn, x.shape[1], x.shape[3], x.shape[4])
        x = x.contiguous(memory_format=torch.channels_last)
        x = self.trunk(x)
        return x.view(B, N, Tn, -1).transpose(2, 3)
