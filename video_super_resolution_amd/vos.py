"""VOSProjectionModule for the gfx950 path (reference my_packages/VOSProjection/*).

OSVOS's VGG trunk runs on stock PyTorch-ROCm convolutions by scope.  The wrapper's
host round trips (numpy mean subtraction, numpy sigmoid/threshold; VOSProjectionModule.py:15-26) are done
on the device instead.  Layer containers keep the reference's names so `parent_epoch-239.pth` keys load.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .trunk_f32 import Conv2dF32, FusedSequential

_STAGES = [[64, 64], ["M", 128, 128], ["M", 256, 256, 256], ["M", 512, 512, 512], ["M", 512, 512, 512]]
_STAGE_IN = [3, 64, 128, 256, 512]
MEANVAL = (104.00699, 116.66877, 122.67892)


def _stage(cfg, cin):
    layers = []
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))
        else:
            layers += [Conv2dF32(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return FusedSequential(*layers)


def _center_crop(x, hh, ww):
    """utils/object_utils.py:6-10 (negative padding): the odd pixel is removed at the right/bottom."""
    dh, dw = x.shape[2] - hh, x.shape[3] - ww
    return x[:, :, dh // 2: x.shape[2] - (dh - dh // 2), dw // 2: x.shape[3] - (dw - dw // 2)]


class OSVOS(nn.Module):
    """vgg_osvos.py:14-62; `forward` returns only the fused logit (the wrapper uses outputs[-1], :20-21)."""

    def __init__(self, pretrained=None):
        super().__init__()
        self.stages = nn.ModuleList(_stage(c, i) for c, i in zip(_STAGES, _STAGE_IN))
        self.side_prep = nn.ModuleList()
        self.score_dsn = nn.ModuleList()
        self.upscale = nn.ModuleList()
        self.upscale_ = nn.ModuleList()
        for i in range(1, len(_STAGES)):
            self.side_prep.append(Conv2dF32(_STAGES[i][-1], 16, kernel_size=3, padding=1))
            self.score_dsn.append(nn.Conv2d(16, 1, kernel_size=1, padding=0))
            self.upscale_.append(nn.ConvTranspose2d(1, 1, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))
            self.upscale.append(nn.ConvTranspose2d(16, 16, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))
        self.fuse = Conv2dF32(64, 1, kernel_size=1, padding=0)

    def forward(self, x):
        hh, ww = x.shape[-2:]
        x = self.stages[0](x)
        side = []
        for i in range(1, len(self.stages)):
            x = self.stages[i](x)
            side.append(_center_crop(self.upscale[i - 1](self.side_prep[i - 1](x)), hh, ww))
        # the per-stage score maps (score_dsn/upscale_) only feed outputs the wrapper discards
        return self.fuse(torch.cat(side, dim=1))


class VOSProjectionModule(nn.Module):
    def __init__(self, pretrained=None):
        super().__init__()
        self.net = OSVOS(pretrained)
        self.register_buffer("meanval", torch.tensor(MEANVAL, dtype=torch.float32), persistent=False)

    @torch.no_grad()
    def forward(self, input1, input2, net=None):
        """[h,w,3] x2 -> [h,w] in {0,1}: sigma(logit_a)+sigma(logit_b) > 0.7 (VOSProjectionModule.py:22-25).
        `net`: optional execution copy of self.net (e.g. float16)."""
        net = net or self.net
        imgs = torch.stack([input1, input2]) - self.meanval  # [2,h,w,3]
        logits = net(imgs.permute(0, 3, 1, 2).contiguous()).float()  # [2,1,h,w]
        s = torch.sigmoid(logits[0, 0]) + torch.sigmoid(logits[1, 0])
        return (s > 0.7).to(torch.float32)
