"""The differentiable evaluation of `SRProjectionModule.forward` for the reference's train step (main.py:205-213: the one call
of `VSR.forward` outside `no_grad`, network/video_super_resolution.py:64) on this repository's own HIP kernels.

torch.autograd is used for what it is here -- the walk over the graph, parameter `.grad` accumulation, views, `cat` -- and every
VALUE and every GRADIENT comes from csrc/sr_train.hip through the C ABI (include/vsr_hip.h, "Train step"): convolution and
transposed convolution forward (each is also the other's input gradient: the FeedbackBlock's k8 s4 pair is adjoint), weight
gradients as pixel correlations, bias / PReLU-slope gradients as fixed-order reductions, the fusion MLP's backward.  float32,
NCHW, the zero-fill FeedbackBlock of SRProjectionModule.py:44-90 (defect D1): group `idx` sees only slice `idx` of its 1x1
"tran" convolution, fed by the previous group's tensor; group 0 sees zeros.  Gradients are checked against the CPU checker's
autograd in tests/test_gpu_train_step.py; there is no stock-operator fallback (a CPU tensor raises, like everywhere else).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L


def _ws(n_floats: int, dev) -> torch.Tensor:
    return torch.empty(max(int(n_floats), 1), dtype=torch.float32, device=dev)


def _conv2d(x, w_kkio, b, cout, K, s, p):
    N, cin, H, W = x.shape
    Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
    out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    L.check(L.load().vsr_train_conv2d_f32(L.dptr(x), L.dptr(w_kkio), L.optr(b), L.dptr(out), N, cin, H, W, cout, Ho, Wo, K, s, p, L.stream()),
            "train_conv2d")
    return out


def _deconv2d(x, w_kkio, b, cout, K, s, p, out_hw=None):
    N, cin, H, W = x.shape
    Ho, Wo = out_hw or ((H - 1) * s - 2 * p + K, (W - 1) * s - 2 * p + K)
    out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    L.check(L.load().vsr_train_deconv2d_f32(L.dptr(x), L.dptr(w_kkio), L.optr(b), L.dptr(out), N, cin, H, W, cout, Ho, Wo, K, s, p, L.stream()),
            "train_deconv2d")
    return out


def _corr_dw(small, big, K, s, p):
    """dw[a][b][ky][kx] = sum small[n,a,oy,ox] * big[n,b,s oy - p + ky, s ox - p + kx]."""
    N, A, oh, ow = small.shape
    _, Bc, BH, BW = big.shape
    lib = L.load()
    dw = torch.empty((A, Bc, K, K), dtype=torch.float32, device=small.device)
    ws = _ws(lib.vsr_train_corr_dw_ws_floats(N, A, oh, Bc, K), small.device)
    L.check(lib.vsr_train_corr_dw_f32(L.dptr(small), L.dptr(big), L.dptr(dw), L.dptr(ws), N, A, oh, ow, Bc, BH, BW, K, s, p, L.stream()),
            "train_corr_dw")
    return dw


def _chan_sum(g):
    N, C = g.shape[0], g.shape[1]
    P = g[0, 0].numel()
    db = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = _ws(N * 16 * C, g.device)
    L.check(L.load().vsr_train_chan_sum_f32(L.dptr(g), L.dptr(db), L.dptr(ws), N, C, ctypes.c_size_t(P), L.stream()), "train_chan_sum")
    return db


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d (blocks.py:16-22): weight [Cout,Cin,K,K]."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, b is not None)
        return _conv2d(x, w.detach().permute(2, 3, 1, 0).contiguous(), None if b is None else b.detach().contiguous(), w.shape[0], w.shape[2],
                       stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s, p, has_b = ctx.cfg
        g = g.contiguous()
        K = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # ConvTranspose2d of the gradient with the weight's roles swapped: [ky][kx][cout][cin]
            dx = _deconv2d(g, w.detach().permute(2, 3, 0, 1).contiguous(), None, w.shape[1], K, s, p, out_hw=(x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            dw = _corr_dw(g, x, K, s, p)
        if has_b and ctx.needs_input_grad[2]:
            db = _chan_sum(g)
        return dx, dw, db, None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d (blocks.py:34): weight [Cin,Cout,K,K]."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, b is not None)
        return _deconv2d(x, w.detach().permute(2, 3, 0, 1).contiguous(), None if b is None else b.detach().contiguous(), w.shape[1], w.shape[2],
                         stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s, p, has_b = ctx.cfg
        g = g.contiguous()
        K = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # Conv2d of the gradient: [ky][kx][cin_k = Cout][cout_k = Cin]
            dx = _conv2d(g, w.detach().permute(2, 3, 1, 0).contiguous(), None, w.shape[0], K, s, p)
        if ctx.needs_input_grad[1]:
            dw = _corr_dw(x, g, K, s, p)
        if has_b and ctx.needs_input_grad[2]:
            db = _chan_sum(g)
        return dx, dw, db, None, None


class PReLUFn(torch.autograd.Function):
    """nn.PReLU(num_parameters=1) (blocks.py:64-71)."""

    @staticmethod
    def forward(ctx, v, a):
        v = v.contiguous()
        slope = a.detach().to(torch.float32).reshape(1).contiguous()   # read by the kernels from device memory: no host sync
        ctx.save_for_backward(v, a, slope)
        y = torch.empty_like(v)
        L.check(L.load().vsr_train_prelu_f32(L.dptr(v), L.dptr(slope), L.dptr(y), ctypes.c_size_t(v.numel()), L.stream()), "train_prelu")
        return y

    @staticmethod
    def backward(ctx, g):
        v, a, slope = ctx.saved_tensors   # `slope`: the value forward used
        g = g.contiguous()
        lib = L.load()
        gv = torch.empty_like(v)
        da = torch.empty(1, dtype=torch.float32, device=v.device)
        ws = _ws(lib.vsr_train_prelu_bwd_ws_floats(ctypes.c_size_t(v.numel())), v.device)
        L.check(lib.vsr_train_prelu_bwd_f32(L.dptr(v), L.dptr(g), L.dptr(slope), L.dptr(gv), L.dptr(da), L.dptr(ws),
                                            ctypes.c_size_t(v.numel()), L.stream()), "train_prelu_bwd")
        return gv, da.reshape(a.shape)


class AffineFn(torch.autograd.Function):
    """y = (a + b) * scale[c] + shift[c] with frozen scale / shift: MeanShift (blocks.py:46-55, `requires_grad = False`) and the
    skip add in front of add_mean (SRProjectionModule.py:142-143).  `b` (the bilinear skip of the input) carries no gradient."""

    @staticmethod
    def forward(ctx, a, b, scale, shift):
        a = a.contiguous()
        ctx.save_for_backward(scale)
        N, C = a.shape[0], a.shape[1]
        y = torch.empty_like(a)
        L.check(L.load().vsr_train_affine_ch_f32(L.dptr(a), L.optr(None if b is None else b.contiguous()), L.dptr(scale), L.optr(shift), L.dptr(y),
                                                 N, C, ctypes.c_size_t(a[0, 0].numel()), L.stream()), "train_affine_ch")
        return y

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = g.contiguous()
        ga = torch.empty_like(g)
        L.check(L.load().vsr_train_affine_ch_f32(L.dptr(g), L.optr(None), L.dptr(scale), L.optr(None), L.dptr(ga), g.shape[0], g.shape[1],
                                                 ctypes.c_size_t(g[0, 0].numel()), L.stream()), "train_affine_ch")
        return ga, None, None, None


class FusionFn(torch.autograd.Function):
    """`fc` over the plane axis (SRProjectionModule.py:126-131,146 with tools.py:118-123's transposes): [P,3,H,W] -> [1,3,H,W]."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2):
        h = h.contiguous()
        ctx.save_for_backward(h, w1, b1, w2, b2)
        planes, _, H, W = h.shape
        out = torch.empty((1, 3, H, W), dtype=torch.float32, device=h.device)
        L.check(L.load().vsr_sr_fc_fuse_f32(L.dptr(h), L.dptr(w1.detach().contiguous()), L.dptr(b1.detach().contiguous()),
                                            L.dptr(w2.detach().reshape(-1).contiguous()), L.dptr(b2.detach().contiguous()), planes, w1.shape[0],
                                            L.dptr(out), H * W, 0, L.stream()), "sr_fc_fuse")
        return out

    @staticmethod
    def backward(ctx, g):
        h, w1, b1, w2, b2 = ctx.saved_tensors
        planes, _, H, W = h.shape
        hid = w1.shape[0]
        Q = 3 * H * W
        g = g.contiguous()
        dev = h.device
        go = torch.empty((1, 1, 3 * H, W), dtype=torch.float32, device=dev)
        gh = torch.empty((1, hid, 3 * H, W), dtype=torch.float32, device=dev)
        rh = torch.empty((1, hid, 3 * H, W), dtype=torch.float32, device=dev)
        dv = torch.empty_like(h)
        L.check(L.load().vsr_train_fc_bwd_f32(L.dptr(h), L.dptr(g), L.dptr(w1.detach().contiguous()), L.dptr(b1.detach().contiguous()),
                                              L.dptr(w2.detach().reshape(-1).contiguous()), L.dptr(b2.detach().contiguous()), planes, hid,
                                              L.dptr(go), L.dptr(gh), L.dptr(rh), L.dptr(dv), ctypes.c_size_t(Q), L.stream()), "train_fc_bwd")
        # parameter gradients = reductions over the Q = 3 H W (channel, pixel) positions; prefc viewed as [1, planes, 3H, W]
        hv = h.reshape(1, planes, 3 * H, W)
        dw1 = _corr_dw(gh, hv, 1, 1, 0).reshape(hid, planes)
        db1 = _chan_sum(gh)
        dw2 = _corr_dw(go, rh, 1, 1, 0).reshape(1, hid)
        db2 = _chan_sum(go)
        return dv, dw1, db1, dw2, db2


def _conv_act(block, x, stride=1, pad=0, w=None):
    conv = block[0]
    y = Conv2dFn.apply(x, conv.weight if w is None else w, conv.bias, stride, pad)
    return PReLUFn.apply(y, block[1].weight) if len(block) > 1 else y


def _deconv_act(block, x, stride, pad):
    dc = block[0]
    return PReLUFn.apply(ConvTranspose2dFn.apply(x, dc.weight, dc.bias, stride, pad), block[1].weight)


def forward_train(m, x: torch.Tensor) -> torch.Tensor:
    """SRProjectionModule.forward (SRProjectionModule.py:133-147), differentiable, on the kernels above.  x [P,3,h,w] float32."""
    from .sr import sr_geometry
    if not x.is_cuda:
        raise L.VsrHipError("the train step runs on the GPU through hand-written HIP kernels; there is no CPU fallback")
    k, st, pd = sr_geometry(m.upscale_factor)
    S = m.upscale_factor
    b = m.block
    nf, G = m.num_features, b.num_groups
    sub_s, sub_b = m._diag(m.sub_mean)
    add_s, add_b = m._diag(m.add_mean)
    x = x.detach().float().contiguous()
    N, _, h, w = x.shape
    x0 = AffineFn.apply(x, None, sub_s, sub_b)                                    # :135 sub_mean
    inter = torch.empty((N, 3, S * h, S * w), dtype=torch.float32, device=x.device)
    L.check(L.load().vsr_train_bilinear_up_f32(L.dptr(x0), L.dptr(inter), N * 3, h, w, S, L.stream()), "train_bilinear_up")   # :136
    f = _conv_act(m.feat_in, _conv_act(m.conv_in, x0, 1, 1))                      # :137-138
    last = f
    hfin = None

    def tran(block, src, idx):   # 1x1 over a [N, nf (idx+1), ...] map that is zero except channel slice idx = src (D1)
        return _conv_act(block, src, w=block[0].weight[:, nf * idx:nf * (idx + 1)])

    for step in range(m.num_steps):                                               # :140
        lr = [_conv_act(b.compress_in, torch.cat((f, last), 1))]                  # :49-53
        hr = []
        for idx in range(G):                                                      # :54-80
            ld_l = torch.zeros_like(lr[0]) if idx == 0 else tran(b.uptranBlocks[idx - 1], lr[idx - 1], idx)
            hr.append(_deconv_act(b.upBlocks[idx], ld_l, st, pd))
            ld_h = torch.zeros_like(hr[0]) if idx == 0 else tran(b.downtranBlocks[idx - 1], hr[idx - 1], idx)
            lr.append(_conv_act(b.downBlocks[idx], ld_h, st, pd))
        last = _conv_act(b.compress_out, torch.cat(lr[1:], 1))                    # :87-89
        if step == m.num_steps - 1:   # (:145 keeps the last step's frame only; the earlier ones have no consumer)
            c = _conv_act(m.conv_out, _deconv_act(m.out, last, st, pd), 1, 1)     # :142
            hfin = AffineFn.apply(c, inter, add_s, add_b)                         # :142-143 inter_res + ..., add_mean
    return FusionFn.apply(hfin, m.fc[0].weight, m.fc[0].bias, m.fc[2].weight, m.fc[2].bias)   # :146
