"""Executors of the guidance trunks on the hand-written MFMA convolution (igemm.HConv): the fp16 configuration.

Each executor is built from the float32 master module (whose parameters keep the reference's state_dict keys),
packs every convolution once (eval-mode BatchNorm folded in), and evaluates the same graph as the master's
`forward` on NHWC float16 tensors.  Channel concatenations are never materialised by a copy: producers write into
channel slices of the destination buffer.

  HourglassExec  <- depth.build_hourglass()            (reference pytorch_DIW_scratch.py:34-837)
  FlowNet2Exec   <- flownet.FlowNet2                   (reference models.py:73-128, networks/FlowNet*.py)
  OSVOSExec      <- vos.OSVOS                          (reference vgg_osvos.py:47-62)
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .depth import HOURGLASS
from .igemm import (ACT_LEAKY, ACT_NONE, ACT_RELU, HConv, HConvPairS2, HConvStem, HDeconv4s2, HFlowHead, HHourglassFront, SegMap, cached_zeros,
                    pad32, pool2x2, resize_add, to_nhwc_half)


def _fold(conv: nn.Conv2d, bn):
    """(weight, bias) of conv followed by eval-mode BatchNorm (or conv alone when bn is None)."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        scale = torch.rsqrt(bn.running_var.float() + bn.eps)
        if bn.affine:
            scale = scale * bn.weight.detach().float()
        shift = -bn.running_mean.float() * scale
        if bn.affine:
            shift = shift + bn.bias.detach().float()
        w = w * scale.view(-1, 1, 1, 1)
        b = b * scale + shift
    return w, b


def _nchw(x):  # NHWC tensor -> NCHW view (channels_last memory) for stock pooling / resize ops
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------------ hourglass
class _Inception:
    """cat(1x1 | 1x1->k1 | 1x1->k2 | 1x1->k3), every conv + BN + ReLU.  One fused launch evaluates all four 1x1s into
    a work buffer laid out [mid1 | mid2 | mid3 | o0 | o1 | o2 | o3]; the three kxk convs read their mid slice and write
    their o slice; the block's result is the channel slice [M, M+Ctot) of that buffer."""

    def __init__(self, mod):  # mod: ChannelConcat of 4 Sequentials
        b0 = mod[0]
        w0, bb0 = _fold(b0[0], b0[1])
        ws, bs, self.kconvs = [], [], []
        mids, outs = [], []
        for j in (1, 2, 3):
            br = mod[j]
            w1, b1 = _fold(br[0], br[1])
            ws.append(w1)
            bs.append(b1)
            mids.append(w1.shape[0])
            wk, bk = _fold(br[3], br[4])
            outs.append(wk.shape[0])
            self.kconvs.append((wk, bk, br[3].kernel_size[0]))
        self.M = sum(mids)
        self.o0 = w0.shape[0]
        self.ctot = self.o0 + sum(outs)
        self.width = pad32(self.M + self.ctot)
        self.first = HConv(torch.cat(ws + [w0], 0), torch.cat(bs + [bb0], 0), act=ACT_RELU)
        packed = []
        moff, ooff = 0, self.M + self.o0
        for (wk, bk, k), mid, o in zip(self.kconvs, mids, outs):
            packed.append((HConv(wk, bk, pad=(k - 1) // 2, act=ACT_RELU), moff, ooff))
            moff += mid
            ooff += o
        self.kconvs = packed
        self.cin = w0.shape[1]

    # 16-channel branches write DENSE [N,H,W,16] maps (a 32-byte slice of a 512-byte pixel row is a partial-line store per
    # pixel: 3x3 64 -> 16 at 4 x 540 x 960 168 -> 118 us); the block's result is then a SegMap for the level's sum.  False: slices
    # of the work buffer as for the wider blocks (the first build; cross-check).
    dense_thin = os.environ.get("VSR_DENSE_THIN", "1") != "0"

    def branches(self, buf):
        """The three k x k branches on the work buffer `buf` (whose [0, M + o0) the fused 1x1 launch has written) ->
        (map, first channel, channels) of the block's result."""
        if self.dense_thin and all(conv.cout == 16 for conv, _, _ in self.kconvs) and self.o0 == 16:
            N, H, W, _ = buf.shape
            outs = torch.empty((len(self.kconvs), N, H, W, 16), dtype=torch.float16, device=buf.device)
            for j, (conv, moff, _) in enumerate(self.kconvs):
                conv(buf, out=outs[j], out_coff=0, in_coff=moff)
            return SegMap([(buf, self.M)] + [(outs[j], 0) for j in range(len(self.kconvs))]), 0, self.ctot
        for conv, moff, ooff in self.kconvs:
            conv(buf, out=buf, out_coff=ooff, in_coff=moff)
        return buf, self.M, self.ctot

    def __call__(self, x, in_coff):
        N, H, W, _ = x.shape
        buf = torch.empty((N, H, W, self.width), dtype=torch.float16, device=x.device)
        if self.width != self.M + self.ctot:
            buf[..., self.M + self.ctot:] = 0
        self.first(x, out=buf, out_coff=0, in_coff=in_coff)
        return self.branches(buf)


class HourglassExec:
    # One launch for the stem + the max pool of the inner arm + the fused 1x1s of the skip arm's inception block
    # (igemm.HHourglassFront, csrc/conv_hg_front.hip) where the program has that shape (the reference's does); False: the three
    # launches of the first build (the cross-check of tests/test_gpu_trunk_exec.py).
    fused_front = os.environ.get("VSR_HG_FRONT", "1") != "0"

    def __init__(self, netg: nn.Sequential):
        self.prog = self._build(HOURGLASS, netg)
        self.front = self._match_front(self.prog)

    @staticmethod
    def _match_front(prog):
        """("S", [stem, ("S", [("M", [("S", ["max", ...]), ("S", [("I", J)])]), "+"]), ...]) with J's fused 1x1 over 128 channels
        -> HHourglassFront, else None."""
        try:
            items = prog[1]
            stem, lvl = items[0], items[1]
            fan = lvl[1][0]
            arm_a, arm_b = fan[1]
            ok = (stem[0] == "stem" and stem[2] == 128 and lvl[0] == "S" and lvl[1][1] == "+" and len(lvl[1]) == 2 and fan[0] == "M" and
                  arm_a[0] == "S" and arm_a[1][0] == "max" and arm_b[0] == "S" and len(arm_b[1]) == 1 and arm_b[1][0][0] == "I" and
                  arm_b[1][0][1].cin == 128)
            return HHourglassFront(stem[1], arm_b[1][0][1].first) if ok else None
        except (IndexError, TypeError, ValueError, AttributeError):
            return None

    def _build(self, node, mod):
        if isinstance(node, str):
            return node
        tag = node[0]
        if tag == "S":
            items, children = [], list(mod)
            i = 0
            while i < len(node[1]):
                ch = node[1][i]
                if isinstance(ch, tuple) and ch[0] == "conv":
                    # conv [+ bn] [+ relu] fused into one launch
                    bn = children[i + 1] if i + 1 < len(children) and isinstance(children[i + 1], nn.BatchNorm2d) else None
                    j = i + (2 if bn is not None else 1)
                    relu = j < len(children) and isinstance(children[j], nn.ReLU)
                    w, b = _fold(children[i], bn)
                    if w.shape[1] <= 4 and w.shape[3] <= 8:   # the 7x7 stem on the RGB frame: dense-K launch on [k,h,w,4]
                        items.append(("stem", HConvStem(w, b, pad=ch[4], act=ACT_RELU if relu else ACT_NONE), ch[2]))
                    else:
                        items.append(("conv", HConv(w, b, pad=ch[4], act=ACT_RELU if relu else ACT_NONE), ch[2]))
                    i = j + (1 if relu else 0)
                    continue
                items.append(self._build(ch, children[i]))
                i += 1
            return ("S", items)
        if tag == "M":
            return ("M", [self._build(c, m) for c, m in zip(node[1], mod)])
        if tag == "I":
            return ("I", _Inception(mod))
        raise ValueError(node)

    def _side(self, dev, level):
        key = (dev.type, dev.index)
        if getattr(self, "_side_key", None) != key:
            self._side_streams, self._side_key = {}, key
        if level not in self._side_streams:
            self._side_streams[level] = torch.cuda.Stream(device=dev)
        return self._side_streams[level]

    def _fan_out(self, subs, x, coff, c, level):
        """The two arms of an hourglass level (the skip inceptions at this resolution / pool -> inner levels -> upsample)
        are independent until their sum: the first runs on a side stream of its own (one per nesting level), so the
        latency-bound low-resolution launches of the inner levels overlap the wide layers of the outer skips."""
        if len(subs) != 2 or not self.concurrent:
            return [self._run(sub, x, coff, c, level + 1) for sub in subs]
        cur = torch.cuda.current_stream(x.device)
        side = self._side(x.device, level)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            first = self._run(subs[0], x, coff, c, level + 1)
            first[0].record_stream(cur)
        x.record_stream(side)
        second = self._run(subs[1], x, coff, c, level + 1)
        cur.wait_stream(side)
        return [first, second]

    # Off by default: alone the trunk gains 8 % (5.5 -> 5.1 ms for four frames) but inside VSR.forward, where FlowNet2
    # already runs beside it, the frame loses 1.8 ms (same-box A/B, tools/depth_ab.py and bench.py); results are identical.
    concurrent = os.environ.get("VSR_HOURGLASS_STREAMS", "0") == "1"
    defer_up = True   # (False: "up" as a pass of its own, the first build -- the cross-check of tests/test_gpu_trunk_exec.py)

    def _run(self, node, x, coff, c, level=0):
        """x: NHWC buffer whose live channels are [coff, coff+c) -> (buffer, coff, c)."""
        if node == "max" or node == "avg":
            return pool2x2(x, coff, c, 0 if node == "max" else 1), 0, c
        if node == "up":
            return resize_add(x, coff, c, (2 * x.shape[1], 2 * x.shape[2])), 0, c
        tag = node[0]
        if tag == "conv":
            if node[2] == 1:   # one out-channel (the trunk's last layer): a dense [N,H,W,1] map, not 2 bytes of a 64-byte pixel row
                N, H, W, _ = x.shape
                Ho, Wo = node[1].out_hw(H, W)
                out = node[1](x, in_coff=coff, out=torch.empty((N, Ho, Wo, 1), dtype=torch.float16, device=x.device))
            else:
                out = node[1](x, in_coff=coff)
            return out, 0, node[2]
        if tag == "stem":
            return node[1](x), 0, node[2]
        if tag == "I":
            buf, m, ctot = node[1](x, coff)
            return buf, m, ctot
        if tag == "S":
            items = node[1]
            for i, ch in enumerate(items):
                if ch == "+":
                    # list from the preceding fan-out; an arm that ends in "up" hands over its low-resolution map (4th field):
                    # UpsamplingNearest2d(2) and the resize-to-b + add run as ONE pass, the x2 map is never written
                    a, ca, na = x[0][:3]
                    b, cb, nb = x[1][:3]
                    a_up, b_up = len(x[0]) > 3, len(x[1]) > 3
                    out_hw = (b.shape[1] << b_up, b.shape[2] << b_up)   # = the second arm's size (AddResized: interpolate(a, b.shape) + b)
                    x, coff, c = resize_add(a, ca, nb, out_hw, b, cb, up2=a_up, b_up2=b_up), 0, nb
                elif isinstance(ch, tuple) and ch[0] == "M":
                    x = self._fan_out(ch[1], x, coff, c, level)
                elif ch == "up" and i == len(items) - 1 and self.defer_up:
                    return x, coff, c, "up2"
                else:
                    x, coff, c = self._run(ch, x, coff, c, level)
            return x, coff, c
        raise ValueError(node)

    @L.on_device
    @torch.no_grad()
    def __call__(self, frames_nhwc3: torch.Tensor) -> torch.Tensor:
        """[k,h,w,3] float frames -> [k,1,h,w] float depth predictions."""
        k, h, w, _ = frames_nhwc3.shape
        stem_first = self.prog[0] == "S" and isinstance(self.prog[1][0], tuple) and self.prog[1][0][0] == "stem"
        x = cached_zeros(self, "in", (k, h, w, 4 if stem_first else 32), frames_nhwc3.device)
        x[..., :3] = frames_nhwc3
        if self.front is not None and self.fused_front and h >= 2 and w >= 2:
            out, coff, c = self._run_fused_front(x)
        else:
            out, coff, c = self._run(self.prog, x, 0, 32)
        return out[..., coff:coff + 1].permute(0, 3, 1, 2).float()

    def _run_fused_front(self, x4):
        """The program with its first three launches (stem, max pool of the inner arm, the skip inception's fused 1x1s) as one."""
        k, h, w, _ = x4.shape
        items = self.prog[1]
        arm_a, arm_b = items[1][1][0][1]
        inc = arm_b[1][0][1]
        buf = torch.empty((k, h, w, inc.width), dtype=torch.float16, device=x4.device)
        if inc.width != inc.M + inc.ctot:
            buf[..., inc.M + inc.ctot:] = 0
        pooled = torch.empty((k, h // 2, w // 2, 128), dtype=torch.float16, device=x4.device)
        self.front(x4, buf, pooled)
        a = self._run(("S", arm_a[1][1:]), pooled, 0, 128, 1)            # the inner arm behind its "max"
        x, coff, c = self._run(("S", ["+"]), [a, inc.branches(buf)], 0, 0)   # the skip arm: the inception's k x k branches
        return self._run(("S", items[2:]), x, coff, c)


# ------------------------------------------------------------------------------------------------ FlowNet2
def _cv(seq, stride=1, act=ACT_LEAKY, cin_pad=None):
    conv = seq[0] if isinstance(seq, nn.Sequential) else seq
    k = conv.kernel_size[0]
    return HConv(conv.weight, conv.bias, stride=conv.stride[0], pad=(k - 1) // 2, act=act, slope=0.1, cin_pad=cin_pad)


class _Refine:
    """Coarse-to-fine decoder shared by FlowNetC/S (predict on the concat) and SD (inter_conv before predict).

    `fused_heads` (default): predict_flow of a level and the flow upsampling into the next level's concat buffer are ONE launch
    (igemm.HFlowHead, csrc/conv_flow_head.hip); False: the generic convolution / transposed convolution per head (the first
    build, kept as the cross-check of tests/test_gpu_trunk_exec.py)."""
    fused_heads = os.environ.get("VSR_FLOW_HEADS", "1") != "0"

    def __init__(self, net, with_inter: bool):
        self.with_inter = with_inter
        self.pred6 = _cv(net.predict_flow6, act=ACT_NONE)
        self.levels = []
        prev_pred = net.predict_flow6
        for lvl, cin_cat in ((5, 1026), (4, 770), (3, 386), (2, 194)):
            dec = getattr(net, f"deconv{lvl}")[0]
            up = getattr(net, f"upsampled_flow{lvl + 1}_to_{lvl}")
            item = dict(deconv=HDeconv4s2(dec.weight, dec.bias, act=ACT_LEAKY),
                        upflow=HDeconv4s2(up.weight, up.bias, act=ACT_NONE),
                        cat=pad32(cin_cat), dec_c=dec.weight.shape[1])
            pred = getattr(net, f"predict_flow{lvl}")
            if with_inter:
                item["inter"] = _cv(getattr(net, f"inter_conv{lvl}"), act=ACT_NONE, cin_pad=pad32(cin_cat))
                item["pred"] = _cv(pred, act=ACT_NONE)
            else:
                item["pred"] = _cv(pred, act=ACT_NONE, cin_pad=pad32(cin_cat))
            # the head that PRODUCES this level's upsampled flow: the previous level's predict_flow + this level's upsampling
            item["head_in"] = HFlowHead(prev_pred.weight, prev_pred.bias, up.weight, up.bias,
                                        cin_pad=None if (with_inter or lvl == 5) else self.levels[-1]["cat"])
            prev_pred = pred
            self.levels.append(item)
        self.head_last = HFlowHead(prev_pred.weight, prev_pred.bias, cin_pad=None if with_inter else self.levels[-1]["cat"])

    def alloc_cat(self, idx, N, H, W, dev):
        """Concat buffer of decoder level idx (0 -> level 5 ...) at its resolution; the encoder writes slice 0."""
        return cached_zeros(self, f"cat{idx}", (N, H, W, self.levels[idx]["cat"]), dev)

    def __call__(self, c6, cats, enc_c):
        """c6: coarsest features; cats[i]: concat buffer of level 5-i already holding the encoder features in
        [0, enc_c[i]).  Returns flow2 [N,H/4,W/4,32] (2 live channels)."""
        if not self.fused_heads:
            flow = self.pred6(c6)
            src = c6
            for i, lv in enumerate(self.levels):
                cat = cats[i]
                lv["deconv"](src, out=cat, out_coff=enc_c[i])
                lv["upflow"](flow, out=cat, out_coff=enc_c[i] + lv["dec_c"])
                flow = lv["pred"](lv["inter"](cat)) if self.with_inter else lv["pred"](cat)
                src = cat
            return flow
        src = head_src = c6
        for i, lv in enumerate(self.levels):
            cat = cats[i]
            lv["deconv"](src, out=cat, out_coff=enc_c[i])
            lv["head_in"](head_src, up_out=cat, up_coff=enc_c[i] + lv["dec_c"])    # flow of the level above + its upsampling into `cat`
            src = cat
            head_src = lv["inter"](cat) if self.with_inter else cat
        return self.head_last(head_src)


class _FlowNetSExec:
    def __init__(self, net, cin):
        c1 = net.conv1[0] if isinstance(net.conv1, nn.Sequential) else net.conv1
        # 12 -> 64, 7x7, stride 2 on the warped-pair concat: as a convolution over pixel pairs (igemm.HConvPairS2)
        self.conv1 = HConvPairS2(c1.weight, c1.bias, pad=(c1.kernel_size[0] - 1) // 2, act=ACT_LEAKY, slope=0.1) \
            if cin <= 16 and c1.stride[0] == 2 else _cv(net.conv1, cin_pad=pad32(cin))
        self.pair_input = isinstance(self.conv1, HConvPairS2)
        self.names = ["conv2", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"]
        self.convs = {n: _cv(getattr(net, n)) for n in self.names}
        self.refine = _Refine(net, with_inter=False)

    def __call__(self, x):
        N, H, W, _ = x.shape
        dev = x.device
        c1 = self.conv1(x)
        h2, w2 = H // 4, W // 4
        cats = [self.refine.alloc_cat(i, N, H >> (5 - i), W >> (5 - i), dev) for i in range(4)]  # levels 5,4,3,2
        # encoder features land directly in their concat buffers: conv2 -> cat2, conv3_1 -> cat3, conv4_1 -> cat4, conv5_1 -> cat5
        self.convs["conv2"](c1, out=cats[3], out_coff=0)
        # (a conv reads the leading channel slice of a wider buffer in place: in_ld = buffer width, cin = its own)
        self.convs["conv3_1"](self.convs["conv3"](cats[3]), out=cats[2], out_coff=0)
        self.convs["conv4_1"](self.convs["conv4"](cats[2]), out=cats[1], out_coff=0)
        self.convs["conv5_1"](self.convs["conv5"](cats[1]), out=cats[0], out_coff=0)
        c6 = self.convs["conv6_1"](self.convs["conv6"](cats[0]))
        return self.refine(c6, cats, [512, 512, 256, 128])


class _FlowNetCExec:
    def __init__(self, net):
        c1 = net.conv1[0] if isinstance(net.conv1, nn.Sequential) else net.conv1
        self.conv1 = HConvStem(c1.weight, c1.bias, stride=c1.stride[0], pad=(c1.kernel_size[0] - 1) // 2, act=ACT_LEAKY, slope=0.1)
        self.conv2, self.conv3 = _cv(net.conv2), _cv(net.conv3)
        self.redir = _cv(net.conv_redir)
        self.conv3_1 = _cv(net.conv3_1, cin_pad=pad32(473))
        self.names = ["conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"]
        self.convs = {n: _cv(getattr(net, n)) for n in self.names}
        self.refine = _Refine(net, with_inter=False)

    def __call__(self, x6, both=None):
        """x6: [B,H,W,32] with the two normalised frames in channels 0-2 and 3-5; `both` [2B,H,W,4]: the same frames as
        4-channel pixels (frame a of every pair, then frame b), as vsr_flownet_prepare_pairs writes them."""
        B, H, W, _ = x6.shape
        dev = x6.device
        if both is None:
            both = cached_zeros(self, "both", (2 * B, H, W, 4), dev)   # 4-channel pixels for the dense-K stem
            both[:B, ..., :3] = x6[..., 0:3]
            both[B:, ..., :3] = x6[..., 3:6]
        c2 = self.conv2(self.conv1(both))           # [2B,H/4,W/4,128]
        c3 = self.conv3(c2)                         # [2B,H/8,W/8,256]
        a3, b3 = c3[:B], c3[B:]                     # contiguous halves of the batched encoder output
        h8, w8 = H // 8, W // 8
        cat31 = cached_zeros(self, "cat31", (B, h8, w8, pad32(473)), dev)
        self.redir(a3, out=cat31, out_coff=0)
        # cost volume + LeakyReLU straight into channels [32, 473) of the concat buffer (MFMA, fp16 NHWC)
        L.check(L.load().vsr_flownetc_corr_nhwc_f16(L.dptr(a3, torch.float16), L.dptr(b3, torch.float16), L.dptr(cat31, torch.float16),
                                                    cat31.shape[3], 32, B, h8, w8, c3.shape[3], L.stream()), "flownetc_corr")
        cats = [self.refine.alloc_cat(i, B, H >> (5 - i), W >> (5 - i), dev) for i in range(4)]
        cats[3][..., :128] = c2[:B]                 # out_conv2a
        self.conv3_1(cat31, out=cats[2], out_coff=0)
        self.convs["conv4_1"](self.convs["conv4"](cats[2]), out=cats[1], out_coff=0)
        self.convs["conv5_1"](self.convs["conv5"](cats[1]), out=cats[0], out_coff=0)
        c6 = self.convs["conv6_1"](self.convs["conv6"](cats[0]))
        return self.refine(c6, cats, [512, 512, 256, 128])


class _FlowNetSDExec:
    def __init__(self, net):
        self.conv0 = _cv(net.conv0, cin_pad=32)
        self.names = ["conv1", "conv1_1", "conv2", "conv2_1", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1",
                      "conv6", "conv6_1"]
        self.convs = {n: _cv(getattr(net, n)) for n in self.names}
        self.refine = _Refine(net, with_inter=True)

    def __call__(self, x6):
        N, H, W, _ = x6.shape
        dev = x6.device
        c = self.convs
        c0 = self.conv0(x6)
        c1 = c["conv1_1"](c["conv1"](c0))
        cats = [self.refine.alloc_cat(i, N, H >> (5 - i), W >> (5 - i), dev) for i in range(4)]
        c["conv2_1"](c["conv2"](c1), out=cats[3], out_coff=0)
        c["conv3_1"](c["conv3"](cats[3]), out=cats[2], out_coff=0)
        c["conv4_1"](c["conv4"](cats[2]), out=cats[1], out_coff=0)
        c["conv5_1"](c["conv5"](cats[1]), out=cats[0], out_coff=0)
        c6 = c["conv6_1"](c["conv6"](cats[0]))
        return self.refine(c6, cats, [512, 512, 256, 128])


class _FusionExec:
    def __init__(self, net):
        self.conv0 = _cv(net.conv0, cin_pad=32)
        self.conv1, self.conv1_1 = _cv(net.conv1), _cv(net.conv1_1)
        self.conv2, self.conv2_1 = _cv(net.conv2), _cv(net.conv2_1)
        self.pred2 = _cv(net.predict_flow2, act=ACT_NONE)
        self.deconv1 = HDeconv4s2(net.deconv1[0].weight, net.deconv1[0].bias, act=ACT_LEAKY)
        self.up21 = HDeconv4s2(net.upsampled_flow2_to_1.weight, net.upsampled_flow2_to_1.bias)
        self.inter1 = _cv(net.inter_conv1, act=ACT_NONE, cin_pad=pad32(162))
        self.pred1 = _cv(net.predict_flow1, act=ACT_NONE)
        self.deconv0 = HDeconv4s2(net.deconv0[0].weight, net.deconv0[0].bias, act=ACT_LEAKY, cin_pad=pad32(162))
        self.up10 = HDeconv4s2(net.upsampled_flow1_to_0.weight, net.upsampled_flow1_to_0.bias)
        self.inter0 = _cv(net.inter_conv0, act=ACT_NONE, cin_pad=pad32(82))
        self.pred0 = _cv(net.predict_flow0, act=ACT_NONE)
        # predict_flow + the flow upsampling of the next level as one launch (igemm.HFlowHead), like _Refine
        self.head2 = HFlowHead(net.predict_flow2.weight, net.predict_flow2.bias, net.upsampled_flow2_to_1.weight, net.upsampled_flow2_to_1.bias)

    def __call__(self, x11):
        N, H, W, _ = x11.shape
        dev = x11.device
        cat0 = cached_zeros(self, "cat0", (N, H, W, pad32(82)), dev)
        cat1 = cached_zeros(self, "cat1", (N, H // 2, W // 2, pad32(162)), dev)
        self.conv0(x11, out=cat0, out_coff=0)                                             # 64
        self.conv1_1(self.conv1(cat0), out=cat1, out_coff=0)                              # 128
        c2 = self.conv2_1(self.conv2(cat1))
        # the fused head (predict_flow as a 1x1 onto 18 tap-channels, its four waves sharing the channel chunks) pays from 4 chunks
        # up: predict_flow2 (128 channels).  The 32-channel heads at 1/2 and full resolution stay on the patch kernel and the
        # four-phase transposed convolution (measured, tools/trunk_layers.sh: 150 vs 38 us for predict_flow0 at 2 x 512 x 960)
        if _Refine.fused_heads:
            self.head2(c2, up_out=cat1, up_coff=160)                                      # flow2 + its upsampling (2)
            self.deconv1(c2, out=cat1, out_coff=128)                                      # 32
        else:
            flow2 = self.pred2(c2)
            self.deconv1(c2, out=cat1, out_coff=128)                                      # 32
            self.up21(flow2, out=cat1, out_coff=160)                                      # 2
        flow1 = self.pred1(self.inter1(cat1))
        self.deconv0(cat1, out=cat0, out_coff=64)                                         # 16
        self.up10(flow1, out=cat0, out_coff=80)                                           # 2
        return self.pred0(self.inter0(cat0))


class FlowNet2Exec:
    def __init__(self, net):
        self.div_flow = net.div_flow
        self.c = _FlowNetCExec(net.flownetc)
        self.s1 = _FlowNetSExec(net.flownets_1, 12)
        self.s2 = _FlowNetSExec(net.flownets_2, 12)
        self.sd = _FlowNetSDExec(net.flownets_d)
        self.fusion = _FusionExec(net.flownetfusion)

    def _sd_stream(self, dev):
        key = (dev.type, dev.index)
        if getattr(self, "_sd_key", None) != key:
            self._sd_side, self._sd_key = torch.cuda.Stream(device=dev), key
        return self._sd_side

    @staticmethod
    def _flow_nchw(t):  # [B,h,w,>=2] half -> [B,2,h,w] float
        return t[..., :2].permute(0, 3, 1, 2).float()

    @L.on_device
    @torch.no_grad()
    def __call__(self, inputs):
        """inputs [B,3,2,H,W] (0..255) -> flow [B,2,H,W] float32; mirrors flownet.FlowNet2.forward."""
        B, _, _, H, W = inputs.shape
        frames = inputs.float().permute(0, 2, 3, 4, 1).reshape(2 * B, H, W, 3).contiguous()   # frame a, frame b of every pair
        out = []
        for b0 in range(0, B, 4):   # (the glue kernels take up to four pairs per launch)
            nb = min(4, B - b0)
            out.append(self.run_pairs(frames[2 * b0:2 * (b0 + nb)], [(2 * i, 2 * i + 1) for i in range(nb)]))
        return self._flow_nchw(out[0] if len(out) == 1 else torch.cat(out, 0))

    @L.on_device
    @torch.no_grad()
    def run_pairs(self, frames, pairs, crop=None):
        """frames [F,h,w,3] float32 (0..255), pairs [(index a, index b)] (at most four), crop (y0, x0, H, W) or None ->
        the fusion network's flow as it leaves the MFMA convolution: [B,H,W,32] half, channels 0,1 live (full resolution).

        FlowNet2.forward (models.py:73-128) with the glue between the sub-networks in four fused launches
        (csrc/flow_ops.hip): prepare_pairs (:74-79), up_warp_concat16 twice (:83-91, :95-103), fusion_input (:106-125)."""
        lib = L.load()
        B = len(pairs)
        F_, h, w, _ = frames.shape
        y0, x0, H, W = crop or (0, 0, h, w)
        dev = frames.device
        frames = frames.contiguous()
        x = torch.empty((B, 6, H, W), dtype=torch.float32, device=dev)
        x6 = torch.empty((B, H, W, 32), dtype=torch.float16, device=dev)
        both = torch.empty((2 * B, H, W, 4), dtype=torch.float16, device=dev)
        ws = torch.empty(B * 128 * 3, dtype=torch.float32, device=dev)
        ia = (ctypes.c_int * B)(*[p[0] for p in pairs])
        ib = (ctypes.c_int * B)(*[p[1] for p in pairs])
        L.check(lib.vsr_flownet_prepare_pairs(L.dptr(frames), F_, h, w, ia, ib, B, y0, x0, H, W, L.dptr(ws), L.dptr(x),
                                              L.dptr(x6, torch.float16), L.dptr(both, torch.float16), L.stream()), "flownet_prepare_pairs")
        if not (self.s1.pair_input and self.s2.pair_input):
            raise NotImplementedError("FlowNetS stems other than 12 -> 64, stride 2 (the reference's)")

        def warp_cat(flow2):
            cat = torch.empty((B, H, W, 16), dtype=torch.float16, device=dev)
            L.check(lib.vsr_flownet_up_warp_concat16_f16(L.dptr(x), L.dptr(flow2, torch.float16), flow2.shape[3], 1, L.cf(self.div_flow),
                                                         L.cf(1.0 / self.div_flow), L.dptr(cat, torch.float16), B, H, W, L.stream()),
                    "up_warp_concat16")
            return cat
        # FlowNetSD depends only on the input pair: it runs on its own HIP stream beside the C -> S1 -> S2 chain (whose
        # 1/16..1/64-resolution layers are latency-bound launches of a few hundred workgroups) and joins at the fusion
        main = torch.cuda.current_stream(dev)
        side = self._sd_stream(dev) if os.environ.get("VSR_FLOWSD_STREAM", "1") != "0" else main   # (A/B switch)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sd2 = self.sd(x6)
            sd2.record_stream(main)
        x6.record_stream(side)
        s1_in = warp_cat(self.c(x6, both))
        s2_in = warp_cat(self.s1(s1_in))
        s22 = self.s2(s2_in)
        main.wait_stream(side)
        x11 = torch.empty((B, H, W, 32), dtype=torch.float16, device=dev)
        L.check(lib.vsr_flownet_fusion_input_f16(L.dptr(x), L.dptr(sd2, torch.float16), sd2.shape[3], L.dptr(s22, torch.float16), s22.shape[3],
                                                 L.cf(self.div_flow), L.dptr(x11, torch.float16), B, H, W, L.stream()), "fusion_input")
        return self.fusion(x11)


# ------------------------------------------------------------------------------------------------ OSVOS
class OSVOSExec:
    def __init__(self, net):
        self.stages = []
        for si, stage in enumerate(net.stages):
            items = []
            for m in stage:
                if isinstance(m, nn.MaxPool2d):
                    items.append("M")
                elif isinstance(m, nn.Conv2d):
                    items.append(HConvStem(m.weight, m.bias, pad=1, act=ACT_RELU) if m.in_channels == 3 else
                                 HConv(m.weight, m.bias, pad=1, act=ACT_RELU))
            self.stages.append(items)
        self.side = [HConv(m.weight, m.bias, pad=1, act=ACT_NONE) for m in net.side_prep]
        # upscale (ConvTranspose2d 16->16, k = 2s, stride s) -> centre crop -> cat -> fuse 1x1 is linear: fold the fuse
        # row into each branch's transposed-conv kernel, weff[ky][kx][ci] (csrc/conv_igemm.hip k_osvos_fuse)
        fw = net.fuse.weight.detach().float().view(-1)
        self.up_s = [m.stride[0] for m in net.upscale]
        self.weff = [torch.einsum("iokl,o->kli", m.weight.detach().float(), fw[16 * b:16 * b + 16]).contiguous().half()
                     for b, m in enumerate(net.upscale)]
        self.fuse_b = float(net.fuse.bias.detach())

    @L.on_device
    @torch.no_grad()
    def __call__(self, x_nchw):
        """[2,3,h,w] mean-subtracted frames -> fused logit [2,1,h,w] float32."""
        N = x_nchw.shape[0]
        hh, ww = x_nchw.shape[-2:]
        x = to_nhwc_half(x_nchw, 4)   # the first VGG conv is a dense-K stem launch
        sides = []
        for si, items in enumerate(self.stages):
            for it in items:
                x = pool2x2(x, 0, x.shape[3], 2) if it == "M" else it(x)   # MaxPool2d(2, 2, ceil_mode=True)
            if si > 0:   # side_prep: 16 channels, as a DENSE [N,h',w',16] map (full-line stores, see _Inception.dense_thin)
                hs, ws = x.shape[1], x.shape[2]
                sides.append(self.side[si - 1](x, out=torch.empty((N, hs, ws, 16), dtype=torch.float16, device=x.device)))
        nb = len(sides)
        out = torch.empty((N, 1, hh, ww), dtype=torch.float32, device=x.device)
        vp = ctypes.c_void_p * nb
        ip = ctypes.c_int * nb
        L.check(L.load().vsr_osvos_fuse_f16(vp(*[t.data_ptr() for t in sides]), ip(*[t.shape[1] for t in sides]),
                                            ip(*[t.shape[2] for t in sides]), sides[0].shape[3], vp(*[t.data_ptr() for t in self.weff]),
                                            ip(*self.up_s), nb, L.cf(self.fuse_b), L.dptr(out), N, hh, ww, L.stream()), "osvos_fuse")
        return out


# ------------------------------------------------------------------------------------------------ VGG16 features (loss)
class VGGFeatExec:
    """`nn.Sequential(*list(vgg16().features)[:31])` (loss_function.py:12-13) on the MFMA convolution: 13 conv3x3 + ReLU (the first
    one as a dense-K stem on the RGB input), five MaxPool2d(2, 2).  [N,3,H,W] float (0..255) -> [N,512,H/32,W/32] float32."""

    def __init__(self, seq):
        self.items = []
        for m in seq:
            if isinstance(m, nn.MaxPool2d):
                self.items.append("M")
            elif isinstance(m, nn.Conv2d):
                self.items.append(HConvStem(m.weight, m.bias, pad=1, act=ACT_RELU) if m.in_channels == 3 else
                                  HConv(m.weight, m.bias, pad=1, act=ACT_RELU))

    @L.on_device
    @torch.no_grad()
    def __call__(self, x_nchw):
        x = to_nhwc_half(x_nchw.float().contiguous(), 4)
        for it in self.items:
            x = pool2x2(x, 0, x.shape[3], 0) if it == "M" else it(x)   # MaxPool2d(2, 2): floor mode
        return x.permute(0, 3, 1, 2).float()


class TrunkExecCache:
    """Version-checked cache of an executor built from a master module."""

    def __init__(self, master: nn.Module, factory):
        self.master, self.factory = master, factory
        self._exec, self._key, self._mods = None, None, None

    def key(self):
        if self._mods is None:
            self._mods = [m for m in self.master.modules() if m._parameters or m._buffers]
        return tuple((t.data_ptr(), t._version) for m in self._mods for d in (m._parameters, m._buffers) for t in d.values() if t is not None)

    def get(self):
        # (address, version) of every parameter and buffer, read through the modules' own dictionaries -- a replaced Parameter object is
        # seen like an in-place update; `nn.Module.parameters()` walks the module tree with name bookkeeping, which cost ~0.4 ms per
        # look-up x 6 per frame.  The LIST of sub-modules is taken once: rebuild the cache object after adding / removing sub-modules.
        key = self.key()
        if self._exec is None or key != self._key:
            self._exec, self._key = self.factory(self.master), key
        return self._exec
