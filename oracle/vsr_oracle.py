"""oracle/vsr_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/__init__.py).

Functional, CPU-only restatement of the reference's per-frame forward
(`VSR.forward`, network/video_super_resolution.py:23-69) and of every
sub-network it calls, written against a flat ``{state_dict key: tensor}``
parameter dictionary.  Dense math uses stock PyTorch CPU ops (the reference's
own arithmetic substrate, SURVEY.md 8(c) "third-party arithmetic"); the three
CUDA extensions go through oracle/native_ops.c; `flow2img` is restated in
numpy float64.

Pinned semantics:
  D1  every byte the reference's FeedbackBlock never writes is ZERO
      (`torch.empty` -> zeros).  The slice-copy loops are restated literally,
      including "the last copy wins", so the oracle does not depend on the
      algebraic reduction the product uses.
  flow2img follows the reference under numpy >= 2 promotion rules (the golden
      vectors were captured with numpy 2.2): `u / maxrad + np.finfo(float).eps`
      is a float64 array.

All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import native

Params = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------------------------
# SRProjection  (my_packages/SRProjection/SRProjectionModule.py, blocks.py)
# ----------------------------------------------------------------------------------------------


def _cb(P: Params, pre: str, x, stride=1, padding=0, act=True):
    """ConvBlock = Sequential(Conv2d, PReLU(1 slope))  -- blocks.py:7-27,58-74."""
    y = F.conv2d(x, P[pre + ".0.weight"], P[pre + ".0.bias"], stride=stride, padding=padding)
    return F.prelu(y, P[pre + ".1.weight"]) if act else y


def _db(P: Params, pre: str, x, stride, padding):
    """DeconvBlock = Sequential(ConvTranspose2d, PReLU)  -- blocks.py:30-43."""
    y = F.conv_transpose2d(x, P[pre + ".0.weight"], P[pre + ".0.bias"], stride=stride, padding=padding)
    return F.prelu(y, P[pre + ".1.weight"])


def sr_geometry(upscale_factor: int):
    """(kernel, stride, padding) of the up / down / `out` (de)convolutions.  The reference hard-wires (8, 4, 2)
    (SRProjectionModule.py:10-12,101-103: the x4 row of SRFBN's table); x2 / x3 are that table's other rows, the
    "scale-2 extension" of SURVEY.md 7-1 / 8(d) -- parity-unpinned by construction (the reference crashes there), and
    differing from the pinned x4 case in these three literals only."""
    return {2: (6, 2, 2), 3: (7, 3, 2), 4: (8, 4, 2)}[upscale_factor]


def feedback_block(P: Params, pre: str, x, last_hidden, num_groups: int, nf: int, taps: Optional[dict] = None,
                   stride: int = 4, padding: int = 2):
    """FeedbackBlock.forward with the zero-fill semantic -- SRProjectionModule.py:44-90."""
    lr: List[torch.Tensor] = [_cb(P, pre + "compress_in", torch.cat((x, last_hidden), 1))]  # :49-53
    hr: List[torch.Tensor] = []
    for idx in range(num_groups):  # :54
        shp = list(lr[0].shape)
        shp[1] *= idx + 1
        ld_l = torch.zeros(shp)  # :57 torch.empty -> zeros (D1)
        for i in range(idx):  # :58-59 same slice every time: last one wins
            ld_l[:, nf * idx:nf * (idx + 1)] = lr[i]
        if idx > 0:
            ld_l = _cb(P, f"{pre}uptranBlocks.{idx - 1}", ld_l)  # :62-63
        ld_h = _db(P, f"{pre}upBlocks.{idx}", ld_l, stride, padding)  # :64
        hr.append(ld_h)
        shp = list(hr[0].shape)
        shp[1] *= idx + 1
        ld_hc = torch.zeros(shp)  # :72 (D1)
        for i in range(idx):  # :73-74
            ld_hc[:, nf * idx:nf * (idx + 1)] = hr[i]
        if idx > 0:
            ld_hc = _cb(P, f"{pre}downtranBlocks.{idx - 1}", ld_hc)  # :77-78
        lr.append(_cb(P, f"{pre}downBlocks.{idx}", ld_hc, stride=stride, padding=padding))  # :79-80
    if taps is not None:
        for i, t in enumerate(lr):
            taps[f"lr{i}"] = t
        for i, t in enumerate(hr):
            taps[f"hr{i}"] = t
    return _cb(P, pre + "compress_out", torch.cat(lr[1:], 1))  # :85-88


def sr_forward(P: Params, x: torch.Tensor, pre: str = "", num_steps: int = 3, num_groups: int = 6,
               upscale_factor: int = 4, taps: Optional[dict] = None, group_taps_step: int = -1) -> torch.Tensor:
    """SRProjectionModule.forward -- SRProjectionModule.py:133-147.  x: [8,3,h,w] -> [1,3,4h,4w]."""
    nf = P[pre + "feat_in.0.weight"].shape[0]
    kernel, stride, padding = sr_geometry(upscale_factor)
    assert P[pre + "out.0.weight"].shape[-1] == kernel, "weights were built for another upscale_factor"
    x = F.conv2d(x, P[pre + "sub_mean.weight"], P[pre + "sub_mean.bias"])  # :135
    inter_res = F.interpolate(x, scale_factor=upscale_factor, mode="bilinear", align_corners=False)  # :136
    if taps is not None:
        taps["sub_mean"] = x
    x = _cb(P, pre + "conv_in", x, padding=1)  # :137
    x = _cb(P, pre + "feat_in", x)  # :138
    if taps is not None:
        taps["feat_in"] = x
    last_hidden = x.clone()  # :45-48 (state reset at every forward, :134)
    h = None
    for step in range(num_steps):  # :140
        gt = {} if (taps is not None and step == (group_taps_step % num_steps)) else None
        hid = feedback_block(P, pre + "block.", x, last_hidden, num_groups, nf, gt, stride, padding)
        last_hidden = hid  # :89
        h = inter_res + F.conv2d(_db(P, pre + "out", hid, stride, padding), P[pre + "conv_out.0.weight"],
                                 P[pre + "conv_out.0.bias"], padding=1)  # :142
        h = F.conv2d(h, P[pre + "add_mean.weight"], P[pre + "add_mean.bias"])  # :143
        if taps is not None:
            taps[f"block{step}"] = hid
            taps[f"prefc{step}"] = h
            if gt is not None:
                taps.update({f"g_{k}": v for k, v in gt.items()})
    # only the last step's h survives (:145); fc runs over the batch-of-8 axis (:146, tools.py:118-123)
    v = h.permute(1, 2, 3, 0)  # transpose030112: [8,3,H,W] -> [3,H,W,8]
    v = F.relu(F.linear(v, P[pre + "fc.0.weight"], P[pre + "fc.0.bias"]))
    v = F.relu(F.linear(v, P[pre + "fc.2.weight"], P[pre + "fc.2.bias"]))  # [3,H,W,1]
    out = v.permute(3, 0, 1, 2).squeeze()  # transpose031323 + squeeze -> [3,H,W]
    return out.unsqueeze(0)  # torch.stack over the single surviving step


# ----------------------------------------------------------------------------------------------
# flow2img  (utils/flow_utils.py:4-112), numpy float64
# ----------------------------------------------------------------------------------------------


def middlebury_wheel() -> np.ndarray:
    """55x3 colour wheel -- flow_utils.py:65-112 (segment lengths 15,6,4,11,13,6)."""
    segs = [(15, (0, 1), False), (6, (1, 0), True), (4, (1, 2), False),
            (11, (2, 1), True), (13, (2, 0), False), (6, (0, 2), True)]
    rows = []
    for n, (full, ramp), falling in segs:
        r = np.floor(255.0 * np.arange(n) / n)
        blk = np.zeros((n, 3))
        blk[:, full] = 255.0
        blk[:, ramp] = 255.0 - r if falling else r
        rows.append(blk)
    return np.concatenate(rows, 0)


def flow2img(flow_hw2: np.ndarray) -> np.ndarray:
    """[h,w,2] float32 -> [h,w,3] uint8 -- flow_utils.py:4-62."""
    u = np.array(flow_hw2[:, :, 0], dtype=np.float32)
    v = np.array(flow_hw2[:, :, 1], dtype=np.float32)
    unknown = (np.abs(u) > 1e7) | (np.abs(v) > 1e7)  # :8-12
    u[unknown] = 0
    v[unknown] = 0
    with np.errstate(all="ignore"):
        rad32 = np.sqrt(u * u + v * v)  # float32, :14
        m = np.max(rad32)
        maxrad = m if m > -1 else -1  # python max(-1, m): NaN -> -1  (:15)
        eps = np.finfo(float).eps
        uu = (u / np.float32(maxrad)).astype(np.float64) + eps  # :16-17 (float32 divide, then f64 under NEP-50)
        vv = (v / np.float32(maxrad)).astype(np.float64) + eps
        nan = np.isnan(uu) | np.isnan(vv)  # :33-34
        uu[nan] = 0
        vv[nan] = 0
        wheel = middlebury_wheel()
        ncols = wheel.shape[0]
        rad = np.sqrt(uu * uu + vv * vv)
        a = np.arctan2(-vv, -uu) / np.pi
        fk = (a + 1) / 2 * (ncols - 1) + 1
        k0 = np.floor(fk).astype(int)
        k1 = k0 + 1
        k1[k1 == ncols + 1] = 1
        f = fk - k0
        img = np.zeros(u.shape + (3,), np.uint8)
        inside = rad <= 1
        for c in range(3):
            col0 = wheel[k0 - 1, c] / 255
            col1 = wheel[k1 - 1, c] / 255
            col = (1 - f) * col0 + f * col1
            col = np.where(inside, 1 - rad * (1 - col), col * 0.75)
            img[:, :, c] = np.uint8(np.floor(255 * col * (1 - nan)))
    img[unknown] = 0  # :21-22
    return img


# ----------------------------------------------------------------------------------------------
# FlowNet2  (my_packages/FlowProjection/models.py + networks/*.py); batchNorm=False everywhere
# ----------------------------------------------------------------------------------------------


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def _fconv(P, name, x, stride=1):
    """submodules.conv without BN: Sequential(Conv2d(pad=(k-1)//2), LeakyReLU(0.1)) -- submodules.py:4-17."""
    w = P[name + ".0.weight"]
    return _lrelu(F.conv2d(x, w, P[name + ".0.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2))


def _fdeconv(P, name, x):
    """submodules.deconv: ConvTranspose2d(k4,s2,p1)+LeakyReLU -- submodules.py:36-41."""
    return _lrelu(F.conv_transpose2d(x, P[name + ".0.weight"], P[name + ".0.bias"], stride=2, padding=1))


def _pred(P, name, x):
    return F.conv2d(x, P[name + ".weight"], P[name + ".bias"], padding=1)  # submodules.py:32-33


def _iconv(P, name, x):
    return F.conv2d(x, P[name + ".0.weight"], P[name + ".0.bias"], padding=1)  # submodules.py:20-30


def _upflow(P, name, x):
    return F.conv_transpose2d(x, P[name + ".weight"], P.get(name + ".bias"), stride=2, padding=1)


def _refine_s(P, pre, c2, c3, c4, c5, c6):
    """Shared decoder of FlowNetC / FlowNetS -- FlowNetC.py:93-113, FlowNetS.py:59-80."""
    flow6 = _pred(P, pre + "predict_flow6", c6)
    cat5 = torch.cat((c5, _fdeconv(P, pre + "deconv5", c6), _upflow(P, pre + "upsampled_flow6_to_5", flow6)), 1)
    flow5 = _pred(P, pre + "predict_flow5", cat5)
    cat4 = torch.cat((c4, _fdeconv(P, pre + "deconv4", cat5), _upflow(P, pre + "upsampled_flow5_to_4", flow5)), 1)
    flow4 = _pred(P, pre + "predict_flow4", cat4)
    cat3 = torch.cat((c3, _fdeconv(P, pre + "deconv3", cat4), _upflow(P, pre + "upsampled_flow4_to_3", flow4)), 1)
    flow3 = _pred(P, pre + "predict_flow3", cat3)
    cat2 = torch.cat((c2, _fdeconv(P, pre + "deconv2", cat3), _upflow(P, pre + "upsampled_flow3_to_2", flow3)), 1)
    return _pred(P, pre + "predict_flow2", cat2)


def flownet_c(P, pre, x):
    """FlowNetC.forward (eval) -- FlowNetC.py:61-118."""
    a1 = _fconv(P, pre + "conv1", x[:, 0:3], 2)
    a2 = _fconv(P, pre + "conv2", a1, 2)
    a3 = _fconv(P, pre + "conv3", a2, 2)
    b3 = _fconv(P, pre + "conv3", _fconv(P, pre + "conv2", _fconv(P, pre + "conv1", x[:, 3:], 2), 2), 2)
    corr = torch.from_numpy(native.correlation(a3.numpy(), b3.numpy(), 20, 1, 20, 1, 2))  # :22,:76
    corr = _lrelu(corr)
    redir = _fconv(P, pre + "conv_redir", a3)
    c3 = _fconv(P, pre + "conv3_1", torch.cat((redir, corr), 1))
    c4 = _fconv(P, pre + "conv4_1", _fconv(P, pre + "conv4", c3, 2))
    c5 = _fconv(P, pre + "conv5_1", _fconv(P, pre + "conv5", c4, 2))
    c6 = _fconv(P, pre + "conv6_1", _fconv(P, pre + "conv6", c5, 2))
    return _refine_s(P, pre, a2, c3, c4, c5, c6)


def flownet_s(P, pre, x):
    """FlowNetS.forward (eval) -- FlowNetS.py:51-85."""
    c1 = _fconv(P, pre + "conv1", x, 2)
    c2 = _fconv(P, pre + "conv2", c1, 2)
    c3 = _fconv(P, pre + "conv3_1", _fconv(P, pre + "conv3", c2, 2))
    c4 = _fconv(P, pre + "conv4_1", _fconv(P, pre + "conv4", c3, 2))
    c5 = _fconv(P, pre + "conv5_1", _fconv(P, pre + "conv5", c4, 2))
    c6 = _fconv(P, pre + "conv6_1", _fconv(P, pre + "conv6", c5, 2))
    return _refine_s(P, pre, c2, c3, c4, c5, c6)


def flownet_sd(P, pre, x):
    """FlowNetSD.forward (eval) -- FlowNetSD.py:60-100."""
    c0 = _fconv(P, pre + "conv0", x)
    c1 = _fconv(P, pre + "conv1_1", _fconv(P, pre + "conv1", c0, 2))
    c2 = _fconv(P, pre + "conv2_1", _fconv(P, pre + "conv2", c1, 2))
    c3 = _fconv(P, pre + "conv3_1", _fconv(P, pre + "conv3", c2, 2))
    c4 = _fconv(P, pre + "conv4_1", _fconv(P, pre + "conv4", c3, 2))
    c5 = _fconv(P, pre + "conv5_1", _fconv(P, pre + "conv5", c4, 2))
    c6 = _fconv(P, pre + "conv6_1", _fconv(P, pre + "conv6", c5, 2))
    flow6 = _pred(P, pre + "predict_flow6", c6)
    cat5 = torch.cat((c5, _fdeconv(P, pre + "deconv5", c6), _upflow(P, pre + "upsampled_flow6_to_5", flow6)), 1)
    flow5 = _pred(P, pre + "predict_flow5", _iconv(P, pre + "inter_conv5", cat5))
    cat4 = torch.cat((c4, _fdeconv(P, pre + "deconv4", cat5), _upflow(P, pre + "upsampled_flow5_to_4", flow5)), 1)
    flow4 = _pred(P, pre + "predict_flow4", _iconv(P, pre + "inter_conv4", cat4))
    cat3 = torch.cat((c3, _fdeconv(P, pre + "deconv3", cat4), _upflow(P, pre + "upsampled_flow4_to_3", flow4)), 1)
    flow3 = _pred(P, pre + "predict_flow3", _iconv(P, pre + "inter_conv3", cat3))
    cat2 = torch.cat((c2, _fdeconv(P, pre + "deconv2", cat3), _upflow(P, pre + "upsampled_flow3_to_2", flow3)), 1)
    return _pred(P, pre + "predict_flow2", _iconv(P, pre + "inter_conv2", cat2))


def flownet_fusion(P, pre, x):
    """FlowNetFusion.forward -- FlowNetFusion.py:42-61."""
    c0 = _fconv(P, pre + "conv0", x)
    c1 = _fconv(P, pre + "conv1_1", _fconv(P, pre + "conv1", c0, 2))
    c2 = _fconv(P, pre + "conv2_1", _fconv(P, pre + "conv2", c1, 2))
    flow2 = _pred(P, pre + "predict_flow2", c2)
    cat1 = torch.cat((c1, _fdeconv(P, pre + "deconv1", c2), _upflow(P, pre + "upsampled_flow2_to_1", flow2)), 1)
    flow1 = _pred(P, pre + "predict_flow1", _iconv(P, pre + "inter_conv1", cat1))
    cat0 = torch.cat((c0, _fdeconv(P, pre + "deconv0", cat1), _upflow(P, pre + "upsampled_flow1_to_0", flow1)), 1)
    return _pred(P, pre + "predict_flow0", _iconv(P, pre + "inter_conv0", cat0))


def _warp(img, flow):
    return torch.from_numpy(native.resample2d(img.contiguous().numpy(), flow.contiguous().numpy(), 1, True))


def _cnorm(x):
    return torch.from_numpy(native.channelnorm(x.contiguous().numpy()))


def flownet2_forward(P: Params, inputs: torch.Tensor, pre: str = "", div_flow: float = 20.0) -> torch.Tensor:
    """FlowNet2.forward -- models.py:73-128.  inputs [B,3,2,H,W] (0..255) -> flow [B,2,H,W]."""
    mean = inputs.contiguous().view(inputs.shape[:2] + (-1,)).mean(-1).view(inputs.shape[:2] + (1, 1, 1))  # :74
    x = (inputs - mean) / 255.0
    x = torch.cat((x[:, :, 0], x[:, :, 1]), 1)  # :77-79
    img0, img1 = x[:, :3], x[:, 3:]
    up_bil = lambda t: F.interpolate(t, scale_factor=4, mode="bilinear")  # nn.Upsample(bilinear), :38,:44
    up_nn = lambda t: F.interpolate(t, scale_factor=4, mode="nearest")  # :53-54

    fc = up_bil(flownet_c(P, pre + "flownetc.", x) * div_flow)  # :82-83
    w1 = _warp(img1, fc)
    cat1 = torch.cat((x, w1, fc / div_flow, _cnorm(img0 - w1)), 1)  # :86-91

    fs1 = up_bil(flownet_s(P, pre + "flownets_1.", cat1) * div_flow)  # :94-95
    w2 = _warp(img1, fs1)
    cat2 = torch.cat((x, w2, fs1 / div_flow, _cnorm(img0 - w2)), 1)  # :98-103

    fs2 = up_nn(flownet_s(P, pre + "flownets_2.", cat2) * div_flow)  # :106-107
    n_fs2 = _cnorm(fs2)
    d_fs2 = _cnorm(img0 - _warp(img1, fs2))  # :110-112

    fsd = up_nn(flownet_sd(P, pre + "flownets_d.", x) / div_flow)  # :115-116 (divided, not multiplied)
    n_fsd = _cnorm(fsd)
    d_fsd = _cnorm(img0 - _warp(img1, fsd))  # :119-121

    cat3 = torch.cat((img0, fsd, fs2, n_fsd, n_fs2, d_fsd, d_fs2), 1)  # :124-125
    return flownet_fusion(P, pre + "flownetfusion.", cat3)


def flow_projection(P: Params, img1: torch.Tensor, img2: torch.Tensor, pre: str = "net.") -> torch.Tensor:
    """FlowProjectionModule.forward -- FlowProjectionModule.py:18-33.  [h,w,3] x2 -> [h',w',3] float (uint8 valued)."""
    h, w = img1.shape[:2]
    th, tw = (h // 64) * 64, (w // 64) * 64
    crop = lambda im: im[(h - th) // 2:(h + th) // 2, (w - tw) // 2:(w + tw) // 2, :]  # tools.py:8-14
    images = torch.stack([crop(img1), crop(img2)])  # [2,h',w',3]
    images = images.permute(3, 0, 1, 2).unsqueeze(0)  # :27-28 -> [1,3,2,h',w']
    flow = flownet2_forward(P, images, pre).squeeze()  # [2,h',w']
    flow = flow.permute(1, 2, 0)  # :31 -> [h',w',2]
    return torch.tensor(flow2img(flow.numpy().copy()), dtype=torch.float32)


# ----------------------------------------------------------------------------------------------
# Depth: MegaDepth hourglass (my_packages/DepthProjection/models/pytorch_DIW_scratch.py:34-837)
# ----------------------------------------------------------------------------------------------
# Architecture as data.  I(cin, o0, (mid,k,out)x3) is the 4-branch inception block
# (1x1 | 1x1->kxk x3, every conv followed by BatchNorm(affine=False)+ReLU, channel concat);
# S = sequential, M = "apply every child to the same input" (LambdaMap), '+' = resize-to-second
# and add (coolAddTensors, :29-31).
_A = ("I", 128, 32, (32, 3, 32), (32, 5, 32), (32, 7, 32))
_B = ("I", 128, 64, (32, 3, 64), (32, 5, 64), (32, 7, 64))
_C = ("I", 256, 64, (32, 3, 64), (32, 5, 64), (32, 7, 64))
_D = ("I", 256, 64, (64, 3, 64), (64, 7, 64), (64, 11, 64))
_E = ("I", 256, 32, (32, 3, 32), (32, 5, 32), (32, 7, 32))
_F = ("I", 128, 32, (64, 3, 32), (64, 7, 32), (64, 11, 32))
_G = ("I", 128, 32, (64, 3, 32), (64, 5, 32), (64, 7, 32))
_H = ("I", 128, 16, (32, 3, 16), (32, 7, 16), (32, 11, 16))
_J = ("I", 128, 16, (64, 3, 16), (64, 7, 16), (64, 11, 16))
_L4 = ("S", [("M", [("S", [_C, _C]), ("S", ["avg", _C, _C, _C, "up"])]), "+"])
_L3 = ("S", [("M", [("S", [_C, _D]), ("S", ["avg", _C, _C, _L4, _C, _D, "up"])]), "+"])
_L2 = ("S", [("M", [("S", ["max", _A, _B, _L3, _C, _E, "up"]), ("S", [_A, _F])]), "+"])
_L1 = ("S", [("M", [("S", ["max", _A, _A, _L2, _G, _H, "up"]), ("S", [_J])]), "+"])
HG_SPEC = ("S", [("conv", 3, 128, 7, 3), ("bn", 128, True), "relu", _L1, ("conv", 64, 1, 3, 1)])


def _bn(P, key, x, affine):
    return F.batch_norm(x, P[key + ".running_mean"], P[key + ".running_var"],
                        P[key + ".weight"] if affine else None, P[key + ".bias"] if affine else None,
                        False, 0.1, 1e-5)


def _hg_run(P, key, node, x):
    if node == "relu":
        return F.relu(x)
    if node == "max":
        return F.max_pool2d(x, 2, 2)
    if node == "avg":
        return F.avg_pool2d(x, 2, 2)
    if node == "up":
        return F.interpolate(x, scale_factor=2, mode="nearest")
    if node == "+":
        a, b = x
        return F.interpolate(a, b.shape[-2:]) + b  # coolAddTensors, :29-31
    tag = node[0]
    if tag == "conv":
        return F.conv2d(x, P[key + ".weight"], P[key + ".bias"], padding=node[4])
    if tag == "bn":
        return _bn(P, key, x, node[2])
    if tag == "S":
        for i, ch in enumerate(node[1]):
            x = _hg_run(P, f"{key}.{i}" if key else str(i), ch, x)
        return x
    if tag == "M":
        return [_hg_run(P, f"{key}.{i}", ch, x) for i, ch in enumerate(node[1])]
    if tag == "I":
        _, cin, o0, *rest = node
        outs = [F.relu(_bn(P, f"{key}.0.1", F.conv2d(x, P[f"{key}.0.0.weight"], P[f"{key}.0.0.bias"]), False))]
        for j, (mid, k, o) in enumerate(rest, start=1):
            t = F.relu(_bn(P, f"{key}.{j}.1", F.conv2d(x, P[f"{key}.{j}.0.weight"], P[f"{key}.{j}.0.bias"]), False))
            t = F.conv2d(t, P[f"{key}.{j}.3.weight"], P[f"{key}.{j}.3.bias"], padding=(k - 1) // 2)
            outs.append(F.relu(_bn(P, f"{key}.{j}.4", t, False)))
        return torch.cat(outs, 1)
    raise ValueError(node)


def hg_forward(P: Params, x: torch.Tensor, pre: str = "") -> torch.Tensor:
    """pytorch_DIW_scratch forward (eval-mode BatchNorm).  [1,3,h,w] -> [1,1,h,w]."""
    return _hg_run(P, pre.rstrip("."), HG_SPEC, x)


def depth_projection(P: Params, frames: torch.Tensor, pre: str = "model.netG.") -> torch.Tensor:
    """DepthProjectionModule.forward -- DepthProjectionModule.py:12-18.  [2,h,w,3] -> [h,w]."""
    x = frames.permute(0, 3, 1, 2)
    d = torch.mean(torch.stack([hg_forward(P, x[0:1], pre), hg_forward(P, x[1:2], pre)]), dim=0)
    return torch.squeeze(d[0])


# ----------------------------------------------------------------------------------------------
# VOS: OSVOS (my_packages/VOSProjection/vgg_osvos.py:14-62) + wrapper (VOSProjectionModule.py:14-27)
# ----------------------------------------------------------------------------------------------
_OSVOS_STAGES = [[64, 64], ["M", 128, 128], ["M", 256, 256, 256], ["M", 512, 512, 512], ["M", 512, 512, 512]]
OSVOS_MEAN = np.array((104.00699, 116.66877, 122.67892), dtype=np.float32)


def _crop_center(x, hh, ww):
    """object_utils.center_crop (:6-10): negative F.pad, the extra pixel comes off the right/bottom."""
    dh, dw = x.shape[2] - hh, x.shape[3] - ww
    return x[:, :, dh // 2: x.shape[2] - (dh - dh // 2), dw // 2: x.shape[3] - (dw - dw // 2)]


def osvos_forward(P: Params, x: torch.Tensor, pre: str = "") -> torch.Tensor:
    """OSVOS.forward, last element only (the fused logit) -- vgg_osvos.py:47-62."""
    hh, ww = x.shape[-2:]
    sides = []
    for si, cfg in enumerate(_OSVOS_STAGES):
        li = 0
        for v in cfg:
            if v == "M":
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)
                li += 1
            else:
                x = F.relu(F.conv2d(x, P[f"{pre}stages.{si}.{li}.weight"], P[f"{pre}stages.{si}.{li}.bias"], padding=1))
                li += 2
        if si > 0:
            s = F.conv2d(x, P[f"{pre}side_prep.{si - 1}.weight"], P[f"{pre}side_prep.{si - 1}.bias"], padding=1)
            up = F.conv_transpose2d(s, P[f"{pre}upscale.{si - 1}.weight"], None, stride=2 ** si)
            sides.append(_crop_center(up, hh, ww))
    return F.conv2d(torch.cat(sides, 1), P[pre + "fuse.weight"], P[pre + "fuse.bias"])


def vos_projection(P: Params, img1: torch.Tensor, img2: torch.Tensor, pre: str = "net.") -> torch.Tensor:
    """VOSProjectionModule.forward -- [h,w,3] x2 -> [h,w] in {0,1}."""
    imgs = np.array([np.subtract(im.numpy(), OSVOS_MEAN) for im in (img1, img2)])
    logits = osvos_forward(P, torch.tensor(imgs.transpose((0, 3, 1, 2))), pre)
    preds = np.transpose(logits.numpy(), (0, 2, 3, 1))
    with np.errstate(over="ignore"):
        preds = [np.squeeze(1 / (1 + np.exp(-p))) for p in preds]
    pred = preds[0] + preds[1]
    return torch.tensor(np.where(pred > 0.7, 1.0, 0.0).astype(np.float32))


# ----------------------------------------------------------------------------------------------
# VSR.forward  (network/video_super_resolution.py:23-69), train=False
# ----------------------------------------------------------------------------------------------


def _nhwc2nchw(x):
    return x.permute(0, 3, 1, 2)  # tools.transpose1323


def vsr_forward(P: Params, data: torch.Tensor, estimated_image: Optional[torch.Tensor],
                high_frames: Optional[torch.Tensor] = None, taps: Optional[dict] = None,
                upscale_factor: int = 4) -> torch.Tensor:
    """One frame.  data [3,h,w,3] (0..255 floats), estimated_image None | [1,4h,4w,3] -> output [1,4h,4w,3]
    (4 -> `upscale_factor` for the scale extension, see sr_geometry)."""
    hw = (data.shape[1], data.shape[2])
    d = data.clone()
    flow_pics = torch.stack([flow_projection(P, d[0], d[1], "FlowModule.net."),
                             flow_projection(P, d[1], d[2], "FlowModule.net.")])  # :28-29
    stack3 = lambda m: torch.stack((m,) * 3)  # tools.maskprocess
    depth = torch.stack([stack3(depth_projection(P, d[0:2], "DepthModule.model.netG.")),
                         stack3(depth_projection(P, d[1:3], "DepthModule.model.netG."))])  # :30-31
    frames = _nhwc2nchw(d)  # :33
    flow_pics = F.interpolate(_nhwc2nchw(flow_pics), hw)  # :35 (nearest)
    est = F.interpolate(_nhwc2nchw(estimated_image), hw) if estimated_image is not None else frames[0:1]  # :37-38
    x8 = torch.cat((frames, flow_pics, depth, est), 0)  # :40
    out1 = sr_forward(P, x8, "model.", upscale_factor=upscale_factor)  # :41
    if taps is not None:
        taps.update(pass1_input=x8, pass1_output=out1)

    trip = torch.stack([est[0], F.interpolate(out1, hw)[0], frames[2]]).permute(0, 2, 3, 1)  # :43-44
    flow_pics = torch.stack([flow_projection(P, trip[0], trip[1], "FlowModule.net."),
                             flow_projection(P, trip[1], trip[2], "FlowModule.net.")])  # :46-47
    depth = torch.stack([stack3(depth_projection(P, trip[0:2], "DepthModule.model.netG.")),
                         stack3(depth_projection(P, trip[1:3], "DepthModule.model.netG."))])  # :49-50
    flow_pics = F.interpolate(_nhwc2nchw(flow_pics), hw)  # :52
    mask = stack3(vos_projection(P, trip[0], trip[1], "VOSModule.net."))  # :54

    mid = trip[1].permute(2, 0, 1)  # tools.transpose1201 -> [3,h,w]
    masked = torch.where(mask != 0, torch.zeros_like(mid), mid).unsqueeze(0)  # :58-60 MaskedArray(...).filled(0)
    x8 = torch.cat((_nhwc2nchw(data), flow_pics, depth, masked), 0)  # :57,:62
    out = sr_forward(P, x8, "model.", upscale_factor=upscale_factor).permute(0, 2, 3, 1)  # :64 transpose1312
    if taps is not None:
        taps.update(pass2_input=x8, vos_mask=mask)
    if high_frames is not None:
        high_frames[1] = out  # :66 (broadcast over the leading 1)
    return out


# ----------------------------------------------------------------------------------------------
# Either side of the path: the dataset windows and main.py's per-item tensor preparation
# ----------------------------------------------------------------------------------------------


def video_windows(imgs, splitvideonum: int = 20):
    """utils/video_utils.py:24-27 -- 3-frame windows, chunked by int(length / splitvideonum) FRAMES."""
    length = len(imgs)
    data = [imgs[i:i + 3] for i in range(len(imgs) - 2)]
    out = []
    for i in range(0, length, int(length / splitvideonum)):
        out.append(data[i:i + int(length / splitvideonum)])
    return out


def make_lr(datas_u8: torch.Tensor, scale: int = 4) -> torch.Tensor:
    """main.py:155-159 MakeDataDatasetToTensor: [T,3,H,W,3] uint8 -> [T,3,H/4,W/4,3] float32 (default-mode nearest)."""
    return torch.stack([F.interpolate(d.type(torch.float32).permute(0, 3, 1, 2), (int(d.shape[1] / scale), int(d.shape[2] / scale)))
                        .permute(0, 2, 3, 1) for d in datas_u8])


def make_target_and_hf(datas_u8: torch.Tensor):
    """main.py:161-167 MakeTargetDatasetToTensor / MakeHFDatasetToTensor."""
    return datas_u8[:, 1:2].type(torch.float32), datas_u8.type(torch.float32)


def frames_to_u8(frames: torch.Tensor) -> torch.Tensor:
    """HR write-out convention of this build (the reference never writes frames): rint (half to even), clamp 0..255, NaN -> 0."""
    return torch.nan_to_num(torch.round(frames), nan=0.0).clamp(0, 255).to(torch.uint8)


# ----------------------------------------------------------------------------------------------
# train=True: VSR.loss_calculate (network/video_super_resolution.py:71-80) and loss_function.py
# ----------------------------------------------------------------------------------------------
_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def vgg16_features31(P: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(*list(vgg16().features)[:31]) -- loss_function.py:12-13, utils/models.py:60-79."""
    idx = 0
    for v in _VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, P[f"{pre}{idx}.weight"], P[f"{pre}{idx}.bias"], padding=1))
            idx += 2
    return x


def tv_loss(x: torch.Tensor) -> torch.Tensor:
    """TVLoss.forward -- loss_function.py:36-44 (tv_loss_weight = 1)."""
    b, c, h, w = x.shape
    h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum()
    w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum()
    return 2 * (h_tv / (c * (h - 1) * w) + w_tv / (c * h * (w - 1))) / b


def sr_loss(P: Params, pre: str, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """SR_loss.forward -- loss_function.py:20-28.  [N,H,W,3] each."""
    o, t = _nhwc2nchw(output), _nhwc2nchw(target)
    perception = F.mse_loss(vgg16_features31(P, pre + "loss_network.", o), vgg16_features31(P, pre + "loss_network.", t))
    return F.mse_loss(o, t) + 0.006 * perception + 2e-8 * tv_loss(o)


def flow_loss(P: Params, pre: str, outputs: torch.Tensor) -> torch.Tensor:
    """Flow_loss.forward -- loss_function.py:57-62."""
    return 0.005 * torch.mean(torch.stack((sr_loss(P, pre + "SR_loss.", outputs[0:1], outputs[1:2]),
                                           sr_loss(P, pre + "SR_loss.", outputs[1:2], outputs[2:3]))))


def loss_calculate(P: Params, target: torch.Tensor, outputs: torch.Tensor, state: dict, taps: dict | None = None) -> torch.Tensor:
    """VSR.loss_calculate -- video_super_resolution.py:71-80.  `state` plays GetObjectsForOBJLoss.mask: the OSVOS mask is
    computed on the first call and reused ever after (loss_function.py:69-74, defect D7).  The masked-array calls are the
    reference's own numpy expressions (:87-92, :98-99), including the [3,H,W] mask applied to [H,W,3] data; BOTH variants pass
    `fill_value=0` (getSRMaskedOutputs :89,:92 and getFlowMaskedOutputs :99).  `taps`, when given, receives the four terms
    and the masked tensors (what fixture g10 pins one by one)."""
    if state.get("mask") is None:
        seg = vos_projection(P, outputs[0], outputs[1], "loss4object.VOS.net.")
        state["mask"] = torch.stack((seg == 1,) * 3)
    mask = state["mask"]
    gen_sr = sr_loss(P, "SR_loss.", outputs[0:1], target)
    with np.errstate(all="ignore"):
        masked_output = torch.unsqueeze(torch.tensor(
            np.ma.MaskedArray(outputs[1].numpy().astype(np.uint8), mask, fill_value=0).filled(), dtype=torch.float32), 0)
        masked_target = torch.tensor(np.ma.MaskedArray(target.numpy().astype(np.uint8), mask, fill_value=0).filled(),
                                     dtype=torch.float32)
        obj_sr = sr_loss(P, "SR_loss.", masked_output, masked_target)
        gen_flow = flow_loss(P, "Flow_loss.", outputs)
        masked = torch.stack([torch.tensor(np.ma.MaskedArray(o.numpy().astype(np.uint8), mask, fill_value=0).filled()) for o in outputs]
                             ).type(torch.float32)
        obj_flow = flow_loss(P, "Flow_loss.", masked)
    if taps is not None:
        taps.update(terms=[float(gen_sr), float(obj_sr), float(gen_flow), float(obj_flow)], masked_sr_out=masked_output,
                    masked_sr_tgt=masked_target, masked_flow=masked)
    return gen_sr + obj_sr + 0.006 * gen_flow + 0.006 * obj_flow
