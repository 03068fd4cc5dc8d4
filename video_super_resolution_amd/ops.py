"""FlowNet2's native operators and the flow colour coding on gfx950 (HIP, through the C ABI).

Module names, constructor arguments and call signatures mirror the reference's wrappers so the
FlowNet2 graph reads the same: `Resample2d` (resample2d_package/resample2d.py:42-51), `ChannelNorm`
(channelnorm_package/channelnorm.py:32-39), `Correlation` (correlation_package/correlation.py:50-64).
Forward only: the whole flow branch of `VSR.forward` runs under `torch.no_grad()`
(network/video_super_resolution.py:24).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _lib as L


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


@L.on_device
def resample2d(img: torch.Tensor, flow: torch.Tensor, kernel_size: int = 1, bilinear: bool = True) -> torch.Tensor:
    """out[b,c,y,x] = bilinear(img[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]); indices clamped independently."""
    img, flow = _f32c(img), _f32c(flow)
    B, C, Hi, Wi = img.shape
    Bf, two, H, W = flow.shape
    if two != 2 or Bf != B or (Hi, Wi) != (H, W):
        raise ValueError(f"resample2d: img {tuple(img.shape)} vs flow {tuple(flow.shape)}")
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=img.device)
    L.check(L.load().vsr_resample2d_f32(L.dptr(img), L.dptr(flow), L.dptr(out), B, C, H, W, int(kernel_size),
                                        int(bool(bilinear)), L.stream()), "resample2d")
    return out


@L.on_device
def channelnorm(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x)
    B, C, H, W = x.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    L.check(L.load().vsr_channelnorm_f32(L.dptr(x), L.dptr(out), B, C, H, W, L.stream()), "channelnorm")
    return out


def correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.check(L.load().vsr_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                               ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)), "correlation_out_shape")
    return oc.value, oh.value, ow.value


@L.on_device
def correlation(f1: torch.Tensor, f2: torch.Tensor, pad_size=20, kernel_size=1, max_displacement=20, stride1=1,
                stride2=2, corr_multiply=1) -> torch.Tensor:
    f1, f2 = _f32c(f1), _f32c(f2)
    if f1.shape != f2.shape:
        raise ValueError("correlation: inputs differ in shape")
    B, C, H, W = f1.shape
    oc, oh, ow = correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = torch.empty((B, oc, oh, ow), dtype=torch.float32, device=f1.device)
    L.check(L.load().vsr_correlation_f32(L.dptr(f1), L.dptr(f2), L.dptr(out), B, C, H, W, pad_size, kernel_size,
                                         max_displacement, stride1, stride2, L.stream()), "correlation")
    return out


@L.on_device
def warp_concat(x6: torch.Tensor, flow: torch.Tensor, div_flow: float) -> torch.Tensor:
    """cat(x6, warp(x6[:,3:], flow), flow/div_flow, |x6[:,:3]-warp|) in one kernel (models.py:86-91,98-103)."""
    x6, flow = _f32c(x6), _f32c(flow)
    B, six, H, W = x6.shape
    if six != 6 or tuple(flow.shape) != (B, 2, H, W):
        raise ValueError("warp_concat: bad shapes")
    out = torch.empty((B, 12, H, W), dtype=torch.float32, device=x6.device)
    L.check(L.load().vsr_flownet_warp_concat_f32(L.dptr(x6), L.dptr(flow), L.cf(1.0 / div_flow), L.dptr(out), B, H, W,
                                                 L.stream()), "warp_concat")
    return out


@L.on_device
def warp_norms(x6: torch.Tensor, flow: torch.Tensor):
    """(|flow|, |x6[:,:3] - warp(x6[:,3:], flow)|) without materialising the warp (models.py:107-112,116-121)."""
    x6, flow = _f32c(x6), _f32c(flow)
    B, six, H, W = x6.shape
    if six != 6 or tuple(flow.shape) != (B, 2, H, W):
        raise ValueError("warp_norms: bad shapes")
    nf = torch.empty((B, 1, H, W), dtype=torch.float32, device=x6.device)
    nd = torch.empty_like(nf)
    L.check(L.load().vsr_flownet_warp_norms_f32(L.dptr(x6), L.dptr(flow), L.dptr(nf), L.dptr(nd), B, H, W, L.stream()),
            "warp_norms")
    return nf, nd


@L.on_device
def flow2img(flow_2hw: torch.Tensor) -> torch.Tensor:
    """[2,h,w] float32 flow -> [h,w,3] float32 picture of uint8 values (utils/flow_utils.py:4-62), no host trip."""
    flow = _f32c(flow_2hw)
    two, H, W = flow.shape
    if two != 2:
        raise ValueError("flow2img expects [2,h,w]")
    out = torch.empty((H, W, 3), dtype=torch.float32, device=flow.device)
    ws = torch.empty(4, dtype=torch.int32, device=flow.device)
    L.check(L.load().vsr_flow2img_f32(L.dptr(flow), L.dptr(out), L.dptr(ws, torch.int32), H, W, L.stream()), "flow2img")
    return out


@L.on_device
def flow2img_nhwc(flow_hwc_half: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """[h,w,ld] float16 map whose channels 0,1 are the flow (the fusion network's output as the MFMA convolution leaves it)
    -> [h,w,3] float32 picture; same arithmetic as flow2img (half -> float is exact)."""
    H, W, ld = flow_hwc_half.shape
    if out is None:
        out = torch.empty((H, W, 3), dtype=torch.float32, device=flow_hwc_half.device)
    ws = torch.empty(4, dtype=torch.int32, device=flow_hwc_half.device)
    L.check(L.load().vsr_flow2img_nhwc_f16(L.dptr(flow_hwc_half, torch.float16), ld, L.dptr(out), L.dptr(ws, torch.int32), H, W,
                                           L.stream()), "flow2img_nhwc")
    return out


class Resample2d(nn.Module):
    def __init__(self, kernel_size=1, bilinear=True):
        super().__init__()
        self.kernel_size = kernel_size
        self.bilinear = bilinear

    def forward(self, input1, input2):
        return resample2d(input1, input2, self.kernel_size, self.bilinear)


class ChannelNorm(nn.Module):
    def __init__(self, norm_deg=2):
        super().__init__()
        self.norm_deg = norm_deg  # accepted and ignored, like the reference kernel (channelnorm_kernel.cu:26)

    def forward(self, input1):
        return channelnorm(input1)


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size, self.kernel_size, self.max_displacement = pad_size, kernel_size, max_displacement
        self.stride1, self.stride2, self.corr_multiply = stride1, stride2, corr_multiply

    def forward(self, input1, input2):
        return correlation(input1, input2, self.pad_size, self.kernel_size, self.max_displacement, self.stride1,
                           self.stride2, self.corr_multiply)
