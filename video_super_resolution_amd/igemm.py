"""NHWC fp16 convolution layers on the hand-written MFMA implicit-GEMM kernel (csrc/conv_igemm.hip).

Activations are `[N,H,W,Cp]` float16 tensors whose channel count Cp is padded with zeros to a multiple of 32; a
layer reads a channel slice and writes a channel slice, so `torch.cat` along channels becomes "write into the
destination buffer".  Weights are packed once from the master `nn.Conv2d` / `nn.ConvTranspose2d` parameters.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L

import ctypes

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
_WS_BYTES = 64 << 20
_MAX_SHAPES = 4   # cached_zeros: shapes kept per (owner, tag, device, stream)
_ws = {}


def _splitk_ws(device) -> torch.Tensor:
    """One fp32 scratch buffer per (device, stream) for split-K partial tiles: launches on one stream are ordered, so
    the buffer is reused; trunks running concurrently on different streams must not share it."""
    key = (device.type, device.index, L.raw_stream(device.index))
    if key not in _ws:
        _ws[key] = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
    return _ws[key]


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def _cout_pad(c: int) -> int:
    if c <= 16:
        return 16
    if c <= 32:
        return 32
    return (c + 63) // 64 * 64


def cached_zeros(owner, tag, shape, device) -> torch.Tensor:
    """A zero-initialised fp16 buffer that belongs to `owner` (a layer / executor object) and is handed out again on the
    next call with the same shape on the same stream.  The conv kernels write only the live channel slices, so the
    padding channels stay zero without a fill launch per call (the trunks issued ~100 of those per frame).  Safe because
    every owner produces one such tensor per executor call and its consumers run before the owner's next call."""
    d = owner.__dict__.setdefault("_bufs", {})
    k = (tag, tuple(shape), device.index, L.raw_stream(device.index))
    t = d.pop(k, None)
    if t is None:
        # bounded: a frame alternates between at most a few shapes per owner (batch of 4 / batch of 1, pass 1 / pass 2);
        # beyond _MAX_SHAPES the least recently used shape of this (tag, device, stream) is dropped, so a service that
        # sees many resolutions does not accumulate one buffer set per resolution
        same = [q for q in d if (q[0], q[2], q[3]) == (k[0], k[2], k[3])]
        for q in same[:max(0, len(same) - (_MAX_SHAPES - 1))]:
            del d[q]
        t = torch.zeros(shape, dtype=torch.float16, device=device)
    d[k] = t   # (re)insert at the end: the dict's order is the recency order
    return t


def to_nhwc_half(x_nchw: torch.Tensor, cp: int = None) -> torch.Tensor:
    """[N,C,H,W] any float -> [N,H,W,pad32(C)] float16, zero padded."""
    N, C, H, W = x_nchw.shape
    cp = cp or pad32(C)
    if x_nchw.is_cuda and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous():   # one pass (csrc/conv_igemm.hip k_nchw_to_nhwc_h)
        out = torch.empty((N, H, W, cp), dtype=torch.float16, device=x_nchw.device)
        L.check(L.load().vsr_nchw_f32_to_nhwc_f16(L.dptr(x_nchw), L.dptr(out, torch.float16), N, C, H, W, cp, L.stream()), "nchw_f32_to_nhwc_f16")
        return out
    out = torch.zeros((N, H, W, cp), dtype=torch.float16, device=x_nchw.device)
    out[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return out


def to_nchw_float(x_nhwc: torch.Tensor, c: int) -> torch.Tensor:
    return x_nhwc[..., :c].permute(0, 3, 1, 2).float()


class HConv:
    """One convolution (+bias +activation) launch.  `weight` [Cout,Cin,kh,kw] float; input slice width = pad32(Cin)
    unless `cin_layout` gives the positions of the Cin real channels inside a wider padded slice."""

    def __init__(self, weight, bias, stride=1, pad=0, act=ACT_NONE, slope=0.1, cin_layout=None, cin_pad=None):
        weight = weight.detach().float()
        cout, cin, kh, kw = weight.shape
        dev = weight.device
        self.cout, self.kh, self.kw, self.stride = cout, kh, kw, stride
        self.pad_y = self.pad_x = pad
        self.act, self.slope = act, slope
        self.cin_pad = cin_pad or pad32(cin if cin_layout is None else int(max(cin_layout)) + 1)
        self.cout_pad = _cout_pad(cout)
        wfull = torch.zeros((self.cout_pad, self.cin_pad, kh, kw), dtype=torch.float32, device=dev)
        if cin_layout is None:
            wfull[:cout, :cin] = weight
        else:
            wfull[:cout, torch.as_tensor(cin_layout, device=dev)] = weight
        # [co][ci][ky][kx] -> [tap][chunk][co][32]
        wp = wfull.view(self.cout_pad, self.cin_pad // 32, 32, kh * kw).permute(3, 1, 0, 2).contiguous()
        self.w = wp.to(torch.float16)
        self.b = None
        if bias is not None:
            self.b = torch.zeros(self.cout_pad, dtype=torch.float32, device=dev)
            self.b[:cout] = bias.detach().float()
        self.oy = (1, 0)
        self.ox = (1, 0)

    def out_hw(self, H, W):
        return (H + 2 * self.pad_y - self.kh) // self.stride + 1, (W + 2 * self.pad_x - self.kw) // self.stride + 1

    def __call__(self, x, out=None, out_coff=0, in_coff=0, out_hw=None, stride_x=0):
        N, H, W, in_ld = x.shape
        Ho, Wo = out_hw or self.out_hw(H, W)
        if out is None:
            out = cached_zeros(self, "out", (N, Ho * self.oy[0], Wo * self.ox[0], pad32(out_coff + self.cout)), x.device) \
                if pad32(out_coff + self.cout) != out_coff + self.cout else \
                torch.empty((N, Ho * self.oy[0], Wo * self.ox[0], out_coff + self.cout), dtype=torch.float16, device=x.device)
        tok = L.TIMER.start(f"conv N{N} {H}x{W} c{self.cin_pad}->{self.cout} k{self.kh}x{self.kw} s{self.stride}"
                            f"{' ph' if self.oy[0] > 1 else ''}") if L.TIMER.enabled else None
        L.check(L.load().vsr_conv2d_nhwc_sx_f16(
            L.dptr(x, torch.float16), in_ld, in_coff, L.dptr(self.w, torch.float16), L.optr(self.b), L.dptr(out, torch.float16),
            out.shape[3], out_coff, N, H, W, self.cin_pad, Ho, Wo, self.cout, self.cout_pad, self.kh, self.kw, self.stride,
            stride_x, self.pad_y, self.pad_x, out.shape[1], out.shape[2], self.oy[0], self.oy[1], self.ox[0], self.ox[1], self.act,
            L.cf(self.slope), L.dptr(_splitk_ws(x.device)), ctypes.c_size_t(_WS_BYTES), L.stream()), "conv2d_nhwc_f16")
        L.TIMER.stop(tok)
        if L.ROUTES.enabled:
            L.ROUTES.note(f"conv N{N} {H}x{W} c{self.cin_pad}->{self.cout} k{self.kh}x{self.kw} s{self.stride}")
        return out


class HConvPairS2:
    """Stride-2 first convolution on a map with <= 16 channels (FlowNetS conv1: 12 -> 64, 7x7): the input [N,H,W,16] is
    viewed as pixel PAIRS [N,H,W/2,32] and the kernel columns are folded into (pair tap, parity): column x_in = 2 ox - pad
    + kx = 2 (ox - P) + (kx + s) with P = ceil(pad/2), s = 2P - pad, so pair tap kx' = (kx + s) >> 1 and parity e =
    (kx + s) & 1 (channel 16 e + c).  The result is a stride-(2,1) convolution with kw' = (kw - 1 + s) // 2 + 1 taps per
    row whose K chunks are full: 7 x 4 x 32 instead of 49 x 32 for the 7x7.  Same products, another summation order."""

    def __init__(self, weight, bias, pad, act=ACT_NONE, slope=0.1):
        weight = weight.detach().float()
        cout, cin, kh, kw = weight.shape
        if cin > 16:
            raise ValueError("HConvPairS2: at most 16 input channels")
        dev = weight.device
        P2 = (pad + 1) // 2
        s = 2 * P2 - pad
        kwp = (kw - 1 + s) // 2 + 1
        w2 = torch.zeros((cout, 32, kh, kwp), dtype=torch.float32, device=dev)
        for kx in range(kw):
            w2[:, 16 * ((kx + s) & 1):16 * ((kx + s) & 1) + cin, :, (kx + s) >> 1] = weight[:, :, :, kx]
        self.inner = HConv(w2, bias, stride=2, pad=0, act=act, slope=slope)
        self.inner.pad_y, self.inner.pad_x = pad, P2
        self.kh, self.kw, self.pad, self.cout = kh, kw, pad, cout

    def __call__(self, x16, out=None, out_coff=0):
        N, H, W, c = x16.shape
        assert c == 16 and W % 2 == 0 and x16.is_contiguous()
        Ho, Wo = (H + 2 * self.pad - self.kh) // 2 + 1, (W + 2 * self.pad - self.kw) // 2 + 1
        return self.inner(x16.view(N, H, W // 2, 32), out=out, out_coff=out_coff, out_hw=(Ho, Wo), stride_x=1)


class HConvStem:
    """First convolution of a trunk on an image with <= 4 channels: input [N,H,W,4] float16 (`to_nhwc_half(x, 4)`), one
    MFMA K chunk per kernel row (k = 4 kx + c) instead of one zero-padded 32-channel chunk per tap."""

    def __init__(self, weight, bias, stride=1, pad=0, act=ACT_NONE, slope=0.1):
        weight = weight.detach().float()
        cout, cin, kh, kw = weight.shape
        if cin > 4 or kw > 8:
            raise ValueError("HConvStem: at most 4 input channels and 8 kernel columns")
        dev = weight.device
        self.cout, self.kh, self.kw, self.stride, self.pad = cout, kh, kw, stride, pad
        self.act, self.slope = act, slope
        self.cout_pad = _cout_pad(cout)
        wp = torch.zeros((kh, self.cout_pad, 8, 4), dtype=torch.float32, device=dev)   # [ky][co][kx][c]
        wp[:, :cout, :kw, :cin] = weight.permute(2, 0, 3, 1)
        self.w = wp.view(kh, self.cout_pad, 32).to(torch.float16).contiguous()
        self.b = None
        if bias is not None:
            self.b = torch.zeros(self.cout_pad, dtype=torch.float32, device=dev)
            self.b[:cout] = bias.detach().float()

    def __call__(self, x4, out=None, out_coff=0):
        N, H, W, c4 = x4.shape
        assert c4 == 4 and x4.dtype == torch.float16
        Ho = (H + 2 * self.pad - self.kh) // self.stride + 1
        Wo = (W + 2 * self.pad - self.kw) // self.stride + 1
        if out is None:
            cp = pad32(out_coff + self.cout)
            out = cached_zeros(self, "out", (N, Ho, Wo, cp), x4.device) if cp != out_coff + self.cout else \
                torch.empty((N, Ho, Wo, cp), dtype=torch.float16, device=x4.device)
        tok = L.TIMER.start(f"conv N{N} {H}x{W} c4->{self.cout} k{self.kh}x{self.kw} s{self.stride} stem") if L.TIMER.enabled else None
        L.check(L.load().vsr_conv2d_stem_f16(L.dptr(x4, torch.float16), L.dptr(self.w, torch.float16), L.optr(self.b),
                                             L.dptr(out, torch.float16), out.shape[3], out_coff, N, H, W, Ho, Wo, self.cout,
                                             self.cout_pad, self.kh, self.kw, self.stride, self.pad, self.pad, self.act,
                                             L.cf(self.slope), L.stream()), "conv2d_stem_f16")
        L.TIMER.stop(tok)
        if L.ROUTES.enabled:
            L.ROUTES.note(f"conv N{N} {H}x{W} c4->{self.cout} k{self.kh}x{self.kw} s{self.stride} stem")
        return out


class HDeconv4s2:
    """ConvTranspose2d(k=4, s=2, p=1) (+bias +activation) as four 2x2-tap phase convolutions.
    weight [Cin,Cout,4,4].  Output row oy = 2y+py gathers input rows y+a+base(py), a in {0,1}:
    py=0: base -1, kernel rows (3,1); py=1: base 0, kernel rows (2,0); same along x."""

    def __init__(self, weight, bias, act=ACT_NONE, slope=0.1, cin_pad=None):
        w = weight.detach().float()  # [Cin,Cout,4,4]
        self.cout = w.shape[1]
        self.phases = []
        kmap = {0: (3, 1), 1: (2, 0)}
        for py in (0, 1):
            for px in (0, 1):
                wk = w[:, :, list(kmap[py]), :][:, :, :, list(kmap[px])]  # [Cin,Cout,2,2]
                c = HConv(wk.permute(1, 0, 2, 3), bias, stride=1, pad=0, act=act, slope=slope, cin_pad=cin_pad)
                c.pad_y, c.pad_x = (1 if py == 0 else 0), (1 if px == 0 else 0)
                c.oy, c.ox = (2, py), (2, px)
                self.phases.append(c)

    def __call__(self, x, out=None, out_coff=0, in_coff=0):
        N, H, W, in_ld = x.shape
        if out is None:
            cp = pad32(out_coff + self.cout)
            out = cached_zeros(self, "out", (N, 2 * H, 2 * W, cp), x.device)
        c0 = self.phases[0]
        tok = L.TIMER.start(f"deconv4s2 N{N} {H}x{W} c{c0.cin_pad}->{self.cout}") if L.TIMER.enabled else None
        wp = (ctypes.c_void_p * 4)(*[c.w.data_ptr() for c in self.phases])
        L.check(L.load().vsr_deconv4s2_nhwc_f16(
            L.dptr(x, torch.float16), in_ld, in_coff, wp, L.optr(c0.b), L.dptr(out, torch.float16), out.shape[3], out_coff, N, H, W,
            c0.cin_pad, self.cout, c0.cout_pad, c0.act, L.cf(c0.slope), L.dptr(_splitk_ws(x.device)), ctypes.c_size_t(_WS_BYTES),
            L.stream()), "deconv4s2_nhwc_f16")
        L.TIMER.stop(tok)
        if L.ROUTES.enabled:
            L.ROUTES.note(f"deconv4s2 N{N} {H}x{W} c{c0.cin_pad}->{self.cout}")
        return out


class HHourglassFront:
    """The front of the depth hourglass as one launch (csrc/conv_hg_front.hip): the 7x7 stem (3 -> 128, BatchNorm folded, ReLU)
    whose map is consumed in place by MaxPool2d(2, 2) and by a 1x1 convolution + ReLU (the fused first launch of the skip arm's
    inception block).  stem: an HConvStem (7x7, 128 out-channels, stride 1, ReLU); one: an HConv (1x1, 128 in, ReLU)."""

    def __init__(self, stem: "HConvStem", one: "HConv"):
        if (stem.kh, stem.kw, stem.stride, stem.pad, stem.cout, stem.cout_pad, stem.act) != (7, 7, 1, 3, 128, 128, ACT_RELU):
            raise ValueError("HHourglassFront: the stem is Conv2d(3, 128, 7, 1, 3) + ReLU")
        if (one.kh, one.kw, one.stride, one.cin_pad, one.act) != (1, 1, 1, 128, ACT_RELU) or one.cout % 8 or one.cout_pad > 256:
            raise ValueError("HHourglassFront: the second stage is a 1x1 convolution + ReLU over the stem's 128 channels")
        self.stem, self.one = stem, one

    def __call__(self, x4, out2, pooled=None, stem_out=None):
        """x4 [N,H,W,4] -> out2 [N,H,W,ld2] channels [0, c2) (written), pooled [N,H//2,W//2,128], stem_out [N,H,W,>=128] (optional)."""
        N, H, W, c4 = x4.shape
        assert c4 == 4 and x4.dtype == torch.float16 and out2.shape[:3] == (N, H, W)
        tok = L.TIMER.start(f"hg_front N{N} {H}x{W} c4->128->{self.one.cout}") if L.TIMER.enabled else None
        L.check(L.load().vsr_hg_front_f16(
            L.dptr(x4, torch.float16), L.dptr(self.stem.w, torch.float16), L.optr(self.stem.b), L.dptr(self.one.w, torch.float16), L.optr(self.one.b),
            self.one.cout, self.one.cout_pad, L.optr(stem_out, torch.float16), stem_out.shape[3] if stem_out is not None else 0,
            L.optr(pooled, torch.float16), L.dptr(out2, torch.float16), out2.shape[3], N, H, W, L.stream()), "hg_front")
        L.TIMER.stop(tok)
        if L.ROUTES.enabled:
            L.ROUTES.note(f"hg_front N{N} {H}x{W} c4->128->{self.one.cout}")
        return out2


class HFlowHead:
    """FlowNet's flow head in one launch (csrc/conv_flow_head.hip): predict_flow (Conv2d(cin, 2, 3, 1, 1), no activation) and,
    when `up` is given, the next level's flow upsampling ConvTranspose2d(2, 2, 4, 2, 1) of the predicted flow, written into a
    channel slice of that level's concat buffer.  pred_weight [2,cin,3,3]; up_weight [2,2,4,4] (ConvTranspose2d layout [in,out,k,k])."""

    def __init__(self, pred_weight, pred_bias, up_weight=None, up_bias=None, cin_pad=None):
        w = pred_weight.detach().float()
        cout, cin, kh, kw = w.shape
        if (cout, kh, kw) != (2, 3, 3):
            raise ValueError("HFlowHead: predict_flow is Conv2d(cin, 2, 3, 1, 1)")
        dev = w.device
        self.cin_pad = cin_pad or pad32(cin)
        wp = torch.zeros((self.cin_pad // 32, 32, 32), dtype=torch.float32, device=dev)       # [chunk][n = tap*2+co][k]
        wfull = torch.zeros((2, self.cin_pad, 3, 3), dtype=torch.float32, device=dev)
        wfull[:, :cin] = w
        # n = (ky*3+kx)*2 + co  <-  wfull[co][chunk*32+k][ky][kx]
        wp[:, :18] = wfull.view(2, self.cin_pad // 32, 32, 9).permute(1, 3, 0, 2).reshape(self.cin_pad // 32, 18, 32)
        self.w = wp.to(torch.float16).contiguous()
        self.b = pred_bias.detach().float().contiguous() if pred_bias is not None else None
        self.up_w = self.up_b = None
        if up_weight is not None:
            if tuple(up_weight.shape) != (2, 2, 4, 4):
                raise ValueError("HFlowHead: the flow upsampling is ConvTranspose2d(2, 2, 4, 2, 1)")
            self.up_w = up_weight.detach().to(torch.float16).float().contiguous()   # (the MFMA path multiplies fp16 operands)
            self.up_b = up_bias.detach().float().contiguous() if up_bias is not None else None

    def __call__(self, x, up_out=None, up_coff=0, in_coff=0):
        """x [N,H,W,ld] -> flow [N,H,W,32] (channels 0, 1 live); with `up_out` [N,2H,2W,ld'] the upsampled flow lands in its
        channels [up_coff, up_coff + 2)."""
        N, H, W, in_ld = x.shape
        flow = cached_zeros(self, "flow", (N, H, W, 32), x.device)
        if (self.up_w is None) != (up_out is None):
            raise ValueError("HFlowHead: `up_out` goes with an upsampling weight")
        tok = L.TIMER.start(f"flow_head N{N} {H}x{W} c{self.cin_pad}") if L.TIMER.enabled else None
        L.check(L.load().vsr_flow_head_f16(
            L.dptr(x, torch.float16), in_ld, in_coff, self.cin_pad, L.dptr(self.w, torch.float16), L.optr(self.b), L.dptr(flow, torch.float16), 32, 0,
            L.optr(self.up_w), L.optr(self.up_b), L.optr(up_out, torch.float16), up_out.shape[3] if up_out is not None else 0, up_coff,
            N, H, W, L.stream()), "flow_head")
        L.TIMER.stop(tok)
        if L.ROUTES.enabled:
            L.ROUTES.note(f"flow_head N{N} {H}x{W} c{self.cin_pad}{' +up' if up_out is not None else ''}")
        return flow


def pool2x2(x, coff, c, mode):
    """2x2 stride-2 max (mode 0) / average (mode 1) pool of channel slice [coff, coff+c) -> dense [N,H/2,W/2,c];
    mode 2: max with ceil_mode -> [N,ceil(H/2),ceil(W/2),c]."""
    N, H, W, ld = x.shape
    ho, wo = ((H + 1) // 2, (W + 1) // 2) if mode == 2 else (H // 2, W // 2)
    out = torch.empty((N, ho, wo, c), dtype=torch.float16, device=x.device)
    L.check(L.load().vsr_pool2x2_nhwc_f16(L.dptr(x, torch.float16), ld, coff, L.dptr(out, torch.float16), N, H, W, c, mode, L.stream()),
            "pool2x2")
    return out


class SegMap:
    """A [N,H,W,c] map kept as up to four channel segments of equal width, each a slice (tensor, first channel) of its own tensor:
    what an inception block with 16-channel branches hands to the level's sum (resize_add) -- its branches write dense
    [N,H,W,16] maps (full-line stores) instead of 32-byte slices of one wide pixel row."""

    def __init__(self, segs):
        self.segs = list(segs)              # [(tensor [N,H,W,ld], coff)]
        self.shape = self.segs[0][0].shape  # (N, H, W, .): the spatial shape is what callers read

    @property
    def device(self):
        return self.segs[0][0].device

    def record_stream(self, stream):
        """torch.Tensor.record_stream for every segment (a SegMap produced on a side stream and read on another)."""
        for t, _ in self.segs:
            t.record_stream(stream)


def _seg_args(t, coff, c):
    """(pointer array, ld array, coff array, nseg, keep-alive) of a plain tensor slice or a SegMap for vsr_resize_add_segs_nhwc_f16."""
    segs = t.segs if isinstance(t, SegMap) else [(t, coff)]
    n = len(segs)
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    for tt, _ in segs:
        if tt.dtype != torch.float16 or not tt.is_contiguous():
            raise L.VsrHipError("resize_add: segments must be dense float16 tensors")
    return vp(*[tt.data_ptr() for tt, _ in segs]), ip(*[tt.shape[3] for tt, _ in segs]), ip(*[co for _, co in segs]), n


def resize_add(a, a_coff, c, out_hw, b=None, b_coff=0, up2=False, b_up2=False):
    """nearest-resize slice [a_coff, +c) of `a` to out_hw and (optionally) add slice [b_coff, +c) of `b` -> dense [N,H,W,c].
    up2 / b_up2: `a` / `b` stands for UpsamplingNearest2d(2) of the tensor passed (not materialised; the index arithmetic of the
    two steps; an upsampled `b` is [N, H/2, W/2, .]).  `a` / `b` may be SegMaps (channel segments in separate tensors)."""
    N, Ha, Wa, a_ld = a.shape
    H, W = out_hw
    out = torch.empty((N, H, W, c), dtype=torch.float16, device=a.device)
    ap, al, ac, an = _seg_args(a, a_coff, c)
    bp, bl, bc, bn = _seg_args(b, b_coff, c) if b is not None else (None, None, None, 0)
    L.check(L.load().vsr_resize_add_segs_nhwc_f16(ap, al, ac, an, Ha, Wa, 1 if up2 else 0, bp, bl, bc, bn, 1 if b_up2 else 0,
                                                  L.dptr(out, torch.float16), N, H, W, c, L.stream()), "resize_add_segs")
    return out
