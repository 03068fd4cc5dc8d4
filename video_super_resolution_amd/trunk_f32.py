"""The guidance trunks' convolutions in the float32 configuration on this repository's own float32 MFMA kernel
(csrc/conv_f32_nchw.hip: v_mfma_f32_32x32x2_f32, float32 in, float32 accumulate) instead of stock MIOpen, which has no
gfx950 find-db in this image and runs these layers at 25-45 TFLOP/s (55 % of a frame of BASELINE config C2).

`Conv2dF32` / `ConvTranspose2dF32` ARE `nn.Conv2d` / `nn.ConvTranspose2d` (same parameters, same state_dict keys, same results
on the CPU and under autograd -- they defer to the stock forward there); for a float32 CUDA tensor outside autograd their forward
is the kernel.  The master modules of FlowNet2 (flownet.py), the hourglass (depth.py), OSVOS (vos.py) are built from them.
reference: my_packages/FlowProjection/networks/submodules.py:4-41, DepthProjection/models/pytorch_DIW_scratch.py:34-837,
VOSProjection/vgg_osvos.py:47-62 (their nn.Conv2d / nn.ConvTranspose2d layers)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib as L

ENABLED = os.environ.get("VSR_TRUNK_F32", "1") != "0"     # False: every layer on the stock operator (A/B switch; tests)


def _ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and
            not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)))


class _Packed:
    """Packed weights [taps][ceil16(cin)][ceil32(cout)] of one layer, rebuilt when the parameter changes."""

    def __init__(self):
        self.key, self.w = None, None

    def get(self, weight: torch.Tensor, make):
        key = (weight.data_ptr(), weight._version, weight.device)
        if key != self.key:
            self.w, self.key = make(), key
        return self.w


def _pack(w_oikk: torch.Tensor) -> torch.Tensor:
    """[Co,C,kh,kw] float32 (contiguous, on the device) -> packed."""
    co, c, kh, kw = w_oikk.shape
    out = torch.empty((kh * kw, (c + 15) // 16 * 16, (co + 31) // 32 * 32), dtype=torch.float32, device=w_oikk.device)
    L.check(L.load().vsr_conv2d_f32_pack(L.dptr(w_oikk), L.dptr(out), co, c, kh, kw, 0, L.stream()), "conv2d_f32_pack")
    return out


def conv2d_packed(x, wp, bias, co, kh, kw, stride, pad_y, pad_x, out=None, out_hw=None, oy=(1, 0), ox=(1, 0)):
    """One launch of the float32 convolution: x [N,C,H,W] -> out [N,co,.,.] (allocated when None: the plain output size)."""
    x = x.contiguous()
    N, C, H, W = x.shape
    Ho, Wo = out_hw or ((H + 2 * pad_y - kh) // stride + 1, (W + 2 * pad_x - kw) // stride + 1)
    if out is None:
        out = torch.empty((N, co, Ho * oy[0], Wo * ox[0]), dtype=torch.float32, device=x.device)
    L.check(L.load().vsr_conv2d_nchw_f32(L.dptr(x), L.dptr(wp), L.optr(bias), L.dptr(out), N, C, H, W, co, Ho, Wo, kh, kw, stride, pad_y, pad_x,
                                         out.shape[2], out.shape[3], oy[0], oy[1], ox[0], ox[1], L.stream()), "conv2d_nchw_f32")
    return out


ROUTE = os.environ.get("VSR_TRUNK_F32_ROUTE", "1") != "0"   # True: the layers the stock operator runs faster stay on it (below)


def _own_pays(N, C, H, W, Co, k, stride) -> bool:
    """Where the own kernel beats the stock operator -- per-layer device times of both at the C2 size, tools/conv_f32_ab.py,
    profiles/r04_conv_f32_ab.txt: NOT on (a) launches of a few dozen workgroups (FlowNet's 1/32 and 1/64-resolution layers: the
    kernel has no split-K; 8 x 15 x 1024 -> 1024: 0.95 vs 0.19 ms), (b) RGB stems (3 input channels padded to a 16-channel K step),
    (c) k x k layers with <= 16 out-channels (half of each 32-row MFMA tile is padding)."""
    if not ROUTE:
        return True
    Ho, Wo = (H + 2 * ((k - 1) // 2) - k) // stride + 1, (W + 2 * ((k - 1) // 2) - k) // stride + 1
    co_pad = (Co + 31) // 32 * 32
    bm = 128 if co_pad % 128 == 0 else (64 if co_pad % 64 == 0 else 32)
    wgs = -(-(N * Ho * Wo) // 128) * (co_pad // bm)
    if wgs < 192 or C <= 4:
        return False
    if Co <= 16 and k >= 3:
        return False
    return True


class Conv2dF32(nn.Conv2d):
    def _conv_forward(self, x, weight, bias):
        if (not _ok(x, weight) or self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != "zeros" or isinstance(self.padding, str) or
                self.stride[0] != self.stride[1] or
                not _own_pays(x.shape[0], x.shape[1], x.shape[2], x.shape[3], weight.shape[0], weight.shape[2], self.stride[0])):
            return super()._conv_forward(x, weight, bias)
        pk = self.__dict__.setdefault("_vsr_pack", _Packed())
        with torch.cuda.device(x.device):
            wp = pk.get(weight, lambda: _pack(weight.detach().contiguous()))
            co, _, kh, kw = weight.shape
            return conv2d_packed(x, wp, None if bias is None else bias.detach(), co, kh, kw, self.stride[0], self.padding[0], self.padding[1])


class ConvTranspose2dF32(nn.ConvTranspose2d):
    """ConvTranspose2d(k=4, s=2, p=1) as its four 2x2-tap phase convolutions (output (2y + py, 2x + px) gathers input rows y - 1 + a
    for py = 0 with kernel rows (3, 1), rows y + a for py = 1 with kernel rows (2, 0); the same along x); anything else: stock."""
    _KMAP = {0: (3, 1), 1: (2, 0)}

    def forward(self, x, output_size=None):
        if (not _ok(x, self.weight) or output_size is not None or self.kernel_size != (4, 4) or self.stride != (2, 2) or self.padding != (1, 1) or
                self.output_padding != (0, 0) or self.groups != 1 or self.dilation != (1, 1) or
                not _own_pays(x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.weight.shape[1], 1, 1)):   # (a phase: one launch of H x W pixels)
            return super().forward(x, output_size)
        pk = self.__dict__.setdefault("_vsr_pack", _Packed())
        w = self.weight.detach()

        def make():
            ph = []
            for py in (0, 1):
                for px in (0, 1):
                    wk = w[:, :, list(self._KMAP[py]), :][:, :, :, list(self._KMAP[px])]      # [Cin,Cout,2,2]
                    ph.append(_pack(wk.permute(1, 0, 2, 3).contiguous()))
            return ph
        with torch.cuda.device(x.device):
            phases = pk.get(self.weight, make)
            N, C, H, W = x.shape
            co = w.shape[1]
            out = torch.empty((N, co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
            b = None if self.bias is None else self.bias.detach()
            for i, wp in enumerate(phases):
                py, px = i >> 1, i & 1
                conv2d_packed(x, wp, b, co, 2, 2, 1, 1 if py == 0 else 0, 1 if px == 0 else 0, out=out, out_hw=(H, W), oy=(2, py), ox=(2, px))
            return out
