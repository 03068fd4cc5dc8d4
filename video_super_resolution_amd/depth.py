"""DepthProjectionModule for the gfx950 path (reference my_packages/DepthProjection/*).

The MegaDepth hourglass trunk runs on stock PyTorch-ROCm convolutions by scope (SURVEY.md 2.1 row 10,
8(f) rank 1).  It is generated from a compact architecture table instead of the reference's 800-line
literal, with the same `nn.Sequential` nesting so the reference checkpoint keys (`netG.<i>.<j>...`,
HG_model.py:16-19) still address the same layers.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import trunk_f32
from .trunk_f32 import Conv2dF32, FusedSequential

# ("I", cin, o0, (mid,k,out) x3): inception block = concat of a 1x1 branch and three 1x1 -> kxk branches,
# every conv followed by BatchNorm(affine=False) + ReLU (pytorch_DIW_scratch.py:42-72 is one instance).
_A = ("I", 128, 32, (32, 3, 32), (32, 5, 32), (32, 7, 32))
_B = ("I", 128, 64, (32, 3, 64), (32, 5, 64), (32, 7, 64))
_C = ("I", 256, 64, (32, 3, 64), (32, 5, 64), (32, 7, 64))
_D = ("I", 256, 64, (64, 3, 64), (64, 7, 64), (64, 11, 64))
_E = ("I", 256, 32, (32, 3, 32), (32, 5, 32), (32, 7, 32))
_F = ("I", 128, 32, (64, 3, 32), (64, 7, 32), (64, 11, 32))
_G = ("I", 128, 32, (64, 3, 32), (64, 5, 32), (64, 7, 32))
_H = ("I", 128, 16, (32, 3, 16), (32, 7, 16), (32, 11, 16))
_J = ("I", 128, 16, (64, 3, 16), (64, 7, 16), (64, 11, 16))
# S = chain, M = fan-out of one input to every child, "+" = resize first to second (nearest) and add.
_L4 = ("S", [("M", [("S", [_C, _C]), ("S", ["avg", _C, _C, _C, "up"])]), "+"])
_L3 = ("S", [("M", [("S", [_C, _D]), ("S", ["avg", _C, _C, _L4, _C, _D, "up"])]), "+"])
_L2 = ("S", [("M", [("S", ["max", _A, _B, _L3, _C, _E, "up"]), ("S", [_A, _F])]), "+"])
_L1 = ("S", [("M", [("S", ["max", _A, _A, _L2, _G, _H, "up"]), ("S", [_J])]), "+"])
HOURGLASS = ("S", [("conv", 3, 128, 7, 3), ("bn", 128, True), "relu", _L1, ("conv", 64, 1, 3, 1)])


class FanOut(nn.Sequential):
    def forward(self, x):
        return [m(x) for m in self]


class ChannelConcat(nn.Sequential):
    def forward(self, x):
        # float32 on the GPU, outside autograd: every branch's last Conv2d -> BatchNorm2d -> ReLU launch writes its channels of
        # the concatenation in place (trunk_f32.FusedSequential); the branches are stride-1 "same" convolutions
        if (trunk_f32.FUSE and trunk_f32.ENABLED and x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad) and
                all(isinstance(m, FusedSequential) for m in self)):
            widths = [[c for c in m if isinstance(c, nn.Conv2d)][-1].out_channels for m in self]
            out = torch.empty((x.shape[0], sum(widths), x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
            off = 0
            for m, wd in zip(self, widths):
                m(x, into=(out, off))
                off += wd
            return out
        return torch.cat([m(x) for m in self], 1)


class AddResized(nn.Module):
    def forward(self, xs):
        a, b = xs
        return F.interpolate(a, b.shape[-2:]) + b  # coolAddTensors, pytorch_DIW_scratch.py:29-31


def _cbr(cin, cout, k):
    return [Conv2dF32(cin, cout, k, 1, (k - 1) // 2), nn.BatchNorm2d(cout, 1e-05, 0.1, False), nn.ReLU()]


def _build(node) -> nn.Module:
    if node == "relu":
        return nn.ReLU()
    if node == "max":
        return nn.MaxPool2d((2, 2), (2, 2))
    if node == "avg":
        return nn.AvgPool2d((2, 2), (2, 2))
    if node == "up":
        return nn.UpsamplingNearest2d(scale_factor=2)
    if node == "+":
        return AddResized()
    tag = node[0]
    if tag == "conv":
        return Conv2dF32(node[1], node[2], node[3], 1, node[4])
    if tag == "bn":
        return nn.BatchNorm2d(node[1]) if node[2] else nn.BatchNorm2d(node[1], 1e-05, 0.1, False)
    if tag == "S":
        return FusedSequential(*[_build(c) for c in node[1]])
    if tag == "M":
        return FanOut(*[_build(c) for c in node[1]])
    if tag == "I":
        _, cin, o0, *rest = node
        return ChannelConcat(FusedSequential(*_cbr(cin, o0, 1)),
                             *[FusedSequential(*(_cbr(cin, mid, 1) + _cbr(mid, out, k))) for (mid, k, out) in rest])
    raise ValueError(node)


def build_hourglass() -> nn.Sequential:
    return _build(HOURGLASS)


class HGModel(nn.Module):
    """HG_model.py:8-24 without the checkpoint load (no weights ship with the reference)."""

    def __init__(self, pretrained=None):
        super().__init__()
        self.netG = build_hourglass()
        if pretrained is not None:
            sd = torch.load(pretrained, map_location="cpu")
            self.netG.load_state_dict({k[7:]: v for k, v in sd.items()})  # strips "module." like HG_model.py:16

    def forward(self, x):
        return self.netG(x)


class DepthProjectionModule(nn.Module):
    """DepthProjectionModule.py:7-18: [2,h,w,3] -> [h,w] = mean of the two single-frame predictions."""

    def __init__(self, pretrained=None):
        super().__init__()
        self.model = HGModel(pretrained)

    @torch.no_grad()
    def predict(self, frame_hw3: torch.Tensor) -> torch.Tensor:
        """One frame [h,w,3] -> raw prediction [1,1,h,w] (lets the caller reuse per-frame results)."""
        return self.model(frame_hw3.permute(2, 0, 1).unsqueeze(0))

    @staticmethod
    def combine(d1: torch.Tensor, d2: torch.Tensor) -> torch.Tensor:
        p = torch.mean(torch.stack([d1, d2]), dim=0)  # :16
        return torch.squeeze(p[0])  # :17

    @torch.no_grad()
    def forward(self, input):
        return self.combine(self.predict(input[0]), self.predict(input[1]))
