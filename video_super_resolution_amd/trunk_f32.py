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
SPATIAL = os.environ.get("VSR_TRUNK_F32_SPATIAL", "1") != "0"   # False: no spatial-reuse kernels (the round-4 flat kernel everywhere; A/B)
FUSE = os.environ.get("VSR_TRUNK_F32_FUSE", "1") != "0"     # False: BatchNorm / ReLU / LeakyReLU / concat as separate stock passes (A/B; tests)

STOCK, FLAT, SPATIAL_K = 0, 1, 2
MIN_TILES, MIN_WGS = 120, 192   # launches smaller than these stay off the spatial / the flat kernel (tests set them to 0)


def _route(N, C, H, W, Co, kh, kw, stride, pad_y, pad_x) -> int:
    """Which implementation serves a layer -- per-layer device times of all three at the C2 size (tools/f32_route_table.py,
    profiles/r04_c2_route_table_after_tuning.txt; tools/conv_f32_ab.py, profiles/r04_conv_f32_spatial_ab.txt):
      * predict_flow-shaped layers (<= 4 out-channels, 3x3, >= 64 input channels, small maps): the flat route, which the library serves
        with its K-sharing head kernel from 256 input channels up (0.06-0.14 ms where the stock operator takes 0.2-1.05);
      * stride-1 k x k layers with <= 16 out-channels (k >= 3; v_mfma_f32_16x16x4_f32, no padded rows), with 32 out-channels (k >= 5),
        the RGB stems (k >= 5), given enough tiles to fill the chip: the spatial-reuse kernels;
      * otherwise the flat kernel, except (a) launches of a few dozen workgroups (FlowNet's 1/32 and 1/64-resolution layers: no
        split-K; 8 x 15 x 1024 -> 1024: 0.95 vs 0.19 ms), (b) the remaining RGB stems, (c) k x k layers with <= 16 out-channels that the
        spatial kernels did not take: the stock operator."""
    if not ROUTE:
        return FLAT
    Ho, Wo = (H + 2 * pad_y - kh) // stride + 1, (W + 2 * pad_x - kw) // stride + 1
    co_pad = (Co + 31) // 32 * 32
    bm = 128 if co_pad % 128 == 0 else (64 if co_pad % 64 == 0 else 32)
    if Co <= 4 and kh * kw >= 9 and C >= 64 and stride == 1 and N * Ho * Wo <= 131072:
        return FLAT     # FlowNet's predict_flow heads (c -> 2, 3x3, up to 1026 channels on 8 x 15 ... 128 x 240 pixels): the K-sharing head kernel
    # spatial-reuse kernels: <= 16 out-channels from 3x3 up (112 vs 41 flat / 50 stock TFLOP/s on 4 x 540 x 960 64 -> 16 11x11), 32
    # out-channels from 5x5 up (83-111 vs 76-79 / 50-98), the RGB stems from 5x5 up (3 -> 128 7x7: 1.28 vs 3.83 / 2.59 ms); with 64+
    # out-channels the flat kernel's 128-pixel blocks win (61 vs 94)
    if SPATIAL and stride == 1 and kh * kw >= 9 and kw <= (16 if Co <= 16 else 33) and (C > 4 or kh * kw >= 25) and \
            (Co <= 16 or ((co_pad == 32 or C <= 4) and kh * kw >= 25) or MIN_TILES == 0):
        thin = Co <= 16
        th = 16 if (thin or bm == 32) else 8
        tiles = N * -(-Ho // th) * -(-Wo // 32) * (1 if thin else co_pad // bm)
        if tiles >= MIN_TILES and kw * 4 * (16 if thin else bm) <= 4096:
            return SPATIAL_K
    wgs = -(-(N * Ho * Wo) // 128) * (co_pad // bm)
    if wgs < MIN_WGS or C <= 4:
        return STOCK
    if Co <= 16 and kh * kw >= 9:
        return STOCK
    return FLAT


def _own_pays(N, C, H, W, Co, k, stride) -> bool:
    return _route(N, C, H, W, Co, k, k, stride, (k - 1) // 2, (k - 1) // 2) != STOCK


def conv2d_fused(x, wp, scale, shift, act, slope, co, kh, kw, stride, pad_y, pad_x, route, out=None, coff=0):
    """One launch: act(conv(x) * scale + shift) into channels [coff, coff + co) of `out` (allocated [N,co,Ho,Wo] when None)."""
    x = x.contiguous()
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad_y - kh) // stride + 1, (W + 2 * pad_x - kw) // stride + 1
    if out is None:
        out = torch.empty((N, co, Ho, Wo), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.shape[0] == N and out.shape[2:] == (Ho, Wo), (out.shape, (N, co, Ho, Wo))
    L.check(L.load().vsr_conv2d_act_nchw_f32(L.dptr(x), L.dptr(wp), L.optr(scale), L.optr(shift), int(act), L.cf(slope), L.dptr(out), out.shape[1], coff,
                                             N, C, H, W, co, kh, kw, stride, pad_y, pad_x, route, L.stream()), "conv2d_act_nchw_f32")
    if L.ROUTES.enabled:   # (the full-size parity tests log which kernel every layer ran on)
        L.ROUTES.note_as(f"conv N{N} {H}x{W} c{C}->{co} k{kh}x{kw} s{stride}",
                         L.ROUTES.last() + (" +bn" if scale is not None else "") + (" ->slice" if out.shape[1] != co else ""))
    return out


def _conv_plain(conv) -> bool:
    return (isinstance(conv, Conv2dF32) and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == "zeros" and
            not isinstance(conv.padding, str) and conv.stride[0] == conv.stride[1])


def _conv_route(conv, x) -> int:
    if not _ok(x, conv.weight) or not _conv_plain(conv):
        return STOCK
    return _route(x.shape[0], x.shape[1], x.shape[2], x.shape[3], conv.weight.shape[0], conv.weight.shape[2], conv.weight.shape[3], conv.stride[0],
                  conv.padding[0], conv.padding[1])


class _Folded:
    """scale / shift of Conv2d bias + eval-mode BatchNorm2d, rebuilt when any of the tensors involved changes."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, conv, bn):
        ts = [conv.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias]
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)
        if key != self.key:
            with torch.no_grad():
                scale = torch.rsqrt(bn.running_var.float() + bn.eps)
                if bn.weight is not None:
                    scale = scale * bn.weight.float()
                shift = -bn.running_mean.float() * scale
                if bn.bias is not None:
                    shift = shift + bn.bias.float()
                if conv.bias is not None:
                    shift = shift + conv.bias.float() * scale
                self.val, self.key = (scale.contiguous(), shift.contiguous()), key
        return self.val


def run_conv_group(conv, bn, act, x, out=None, coff=0):
    """Conv2dF32 [-> BatchNorm2d (eval)] [-> ReLU / LeakyReLU] in one launch; None when the layer is not served by the own kernels."""
    route = _conv_route(conv, x)
    if route == STOCK:
        if L.ROUTES.enabled:
            co, _, kh, kw = conv.weight.shape
            L.ROUTES.note_as(f"conv N{x.shape[0]} {x.shape[2]}x{x.shape[3]} c{x.shape[1]}->{co} k{kh}x{kw} s{conv.stride[0]}", "stock")
        return None
    if bn is not None and (bn.training or bn.running_mean is None):
        return None
    pk = conv.__dict__.setdefault("_vsr_pack", _Packed())
    with torch.cuda.device(x.device):
        wp = pk.get(conv.weight, lambda: _pack(conv.weight.detach().contiguous()))
        if bn is not None:
            scale, shift = conv.__dict__.setdefault("_vsr_fold", _Folded()).get(conv, bn)
        else:
            scale, shift = None, (None if conv.bias is None else conv.bias.detach())
        slope = 0.0 if act is None or isinstance(act, nn.ReLU) else float(act.negative_slope)
        co, _, kh, kw = conv.weight.shape
        return conv2d_fused(x, wp, scale, shift, act is not None, slope, co, kh, kw, conv.stride[0], conv.padding[0], conv.padding[1], route, out=out, coff=coff)


class FusedSequential(nn.Sequential):
    """nn.Sequential (same children, same state_dict keys, same results) whose forward runs every Conv2dF32 -> [BatchNorm2d] ->
    [ReLU | LeakyReLU] run of children as ONE launch of the float32 convolution (folded BatchNorm, activation in the epilogue)
    for a float32 CUDA tensor outside autograd; everything else child by child.  `into` = (tensor [N, ctot, H, W], coff): the last
    child group writes its channels there (the concat buffer of an inception block) instead of into a tensor of its own."""

    def forward(self, x, into=None):
        mods = list(self)
        i, n = 0, len(mods)
        fast = FUSE and ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad)
        while i < n:
            m = mods[i]
            if fast and isinstance(m, Conv2dF32):
                j = i + 1
                bn = mods[j] if j < n and isinstance(mods[j], nn.BatchNorm2d) else None
                j += bn is not None
                act = mods[j] if j < n and isinstance(mods[j], (nn.ReLU, nn.LeakyReLU)) else None
                j += act is not None
                last = j == n and into is not None
                y = run_conv_group(m, bn, act, x, out=into[0] if last else None, coff=into[1] if last else 0)
                if y is not None:
                    if last:
                        return None   # written in place
                    x, i = y, j
                    continue
            x = m(x)
            i += 1
        if into is not None:
            into[0][:, into[1]:into[1] + x.shape[1]].copy_(x)
            return None
        return x


class Conv2dF32(nn.Conv2d):
    def _conv_forward(self, x, weight, bias):
        if weight is self.weight and bias is self.bias:
            y = run_conv_group(self, None, None, x)
            if y is not None:
                return y
        if _ok(x, weight) and not x.is_contiguous():
            # A permuted [h,w,3] frame is channels_last-strided: the stock operator would then produce a channels_last map, and every
            # consumer below (the own kernels read NCHW) would copy the full-resolution 128-channel stem output once per inception
            # branch (4.9 ms each at 4 x 540 x 960: 29 ms of a C2 frame, profiles/r04_c2_kernel_stats_spatial_fused.txt).
            x = x.contiguous()
        return super()._conv_forward(x, weight, bias)


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
