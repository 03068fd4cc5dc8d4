"""FlowProjectionModule / FlowNet2 for the gfx950 path.

The convolution trunks are stock PyTorch-ROCm layers (MIOpen) by scope (SURVEY.md 2.1 rows 3-4, 8(f)
rank 1); everything the reference implements natively or on the host is a hand-written HIP kernel:
`Correlation`, the fused warp/diff/norm/concat stages and the flow colour coding (no host round trip).
Attribute names and layer order follow the reference so its FlowNet2 checkpoint keys load unchanged
(`FlowProjectionModule.py:12-14`): models.py:25-71, networks/FlowNet{C,S,SD,Fusion}.py, submodules.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .trunk_f32 import Conv2dF32, ConvTranspose2dF32, FusedSequential


def conv(in_planes, out_planes, kernel_size=3, stride=1):
    """Conv2d(pad=(k-1)//2, bias) + LeakyReLU(0.1); batchNorm is False everywhere on this path (models.py:28)."""
    return FusedSequential(Conv2dF32(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, bias=True),
                           nn.LeakyReLU(0.1, inplace=True))


def i_conv(in_planes, out_planes):
    return nn.Sequential(Conv2dF32(in_planes, out_planes, 3, 1, 1, bias=True))


def predict_flow(in_planes):
    return Conv2dF32(in_planes, 2, 3, 1, 1, bias=True)


def deconv(in_planes, out_planes):
    return nn.Sequential(ConvTranspose2dF32(in_planes, out_planes, 4, 2, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))


class _Decoder(nn.Module):
    """The coarse-to-fine refinement shared by FlowNetC and FlowNetS (FlowNetC.py:93-113, FlowNetS.py:59-80)."""

    def _make_decoder(self, flow_up_bias: bool):
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(1026)
        self.predict_flow4 = predict_flow(770)
        self.predict_flow3 = predict_flow(386)
        self.predict_flow2 = predict_flow(194)
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, f"upsampled_flow{a}_to_{b}", ConvTranspose2dF32(2, 2, 4, 2, 1, bias=flow_up_bias))

    def _decode(self, c2, c3, c4, c5, c6):
        flow6 = self.predict_flow6(c6)
        cat5 = torch.cat((c5, self.deconv5(c6), self.upsampled_flow6_to_5(flow6)), 1)
        flow5 = self.predict_flow5(cat5)
        cat4 = torch.cat((c4, self.deconv4(cat5), self.upsampled_flow5_to_4(flow5)), 1)
        flow4 = self.predict_flow4(cat4)
        cat3 = torch.cat((c3, self.deconv3(cat4), self.upsampled_flow4_to_3(flow4)), 1)
        flow3 = self.predict_flow3(cat3)
        cat2 = torch.cat((c2, self.deconv2(cat3), self.upsampled_flow3_to_2(flow3)), 1)
        return self.predict_flow2(cat2)


class FlowNetC(_Decoder):
    def __init__(self):
        super().__init__()
        self.conv1 = conv(3, 64, 7, 2)
        self.conv2 = conv(64, 128, 5, 2)
        self.conv3 = conv(128, 256, 5, 2)
        self.conv_redir = conv(256, 32, 1, 1)
        self.corr = ops.Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.conv3_1 = conv(473, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self._make_decoder(flow_up_bias=True)

    def forward(self, x):
        # both frames go through the shared stem as one batch of two (same arithmetic, half the launches)
        B = x.shape[0]
        s = torch.cat((x[:, 0:3], x[:, 3:6]), 0)
        a2b2 = self.conv2(self.conv1(s))
        a3b3 = self.conv3(a2b2)
        a2, a3, b3 = a2b2[:B], a3b3[:B], a3b3[B:]
        corr = F.leaky_relu(self.corr(a3, b3), 0.1).to(a3.dtype)  # the cost volume kernel works in float32
        c3 = self.conv3_1(torch.cat((self.conv_redir(a3), corr), 1))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._decode(a2, c3, c4, c5, c6)


class FlowNetS(_Decoder):
    def __init__(self, input_channels=12):
        super().__init__()
        self.conv1 = conv(input_channels, 64, 7, 2)
        self.conv2 = conv(64, 128, 5, 2)
        self.conv3 = conv(128, 256, 5, 2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self._make_decoder(flow_up_bias=False)

    def forward(self, x):
        c2 = self.conv2(self.conv1(x))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return self._decode(c2, c3, c4, c5, c6)


class FlowNetSD(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = conv(6, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.conv3 = conv(128, 256, stride=2)
        self.conv3_1 = conv(256, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.inter_conv5 = i_conv(1026, 512)
        self.inter_conv4 = i_conv(770, 256)
        self.inter_conv3 = i_conv(386, 128)
        self.inter_conv2 = i_conv(194, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(512)
        self.predict_flow4 = predict_flow(256)
        self.predict_flow3 = predict_flow(128)
        self.predict_flow2 = predict_flow(64)
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, f"upsampled_flow{a}_to_{b}", ConvTranspose2dF32(2, 2, 4, 2, 1))

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flow6 = self.predict_flow6(c6)
        cat5 = torch.cat((c5, self.deconv5(c6), self.upsampled_flow6_to_5(flow6)), 1)
        flow5 = self.predict_flow5(self.inter_conv5(cat5))
        cat4 = torch.cat((c4, self.deconv4(cat5), self.upsampled_flow5_to_4(flow5)), 1)
        flow4 = self.predict_flow4(self.inter_conv4(cat4))
        cat3 = torch.cat((c3, self.deconv3(cat4), self.upsampled_flow4_to_3(flow4)), 1)
        flow3 = self.predict_flow3(self.inter_conv3(cat3))
        cat2 = torch.cat((c2, self.deconv2(cat3), self.upsampled_flow3_to_2(flow3)), 1)
        return self.predict_flow2(self.inter_conv2(cat2))


class FlowNetFusion(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = conv(11, 64)
        self.conv1 = conv(64, 64, stride=2)
        self.conv1_1 = conv(64, 128)
        self.conv2 = conv(128, 128, stride=2)
        self.conv2_1 = conv(128, 128)
        self.deconv1 = deconv(128, 32)
        self.deconv0 = deconv(162, 16)
        self.inter_conv1 = i_conv(162, 32)
        self.inter_conv0 = i_conv(82, 16)
        self.predict_flow2 = predict_flow(128)
        self.predict_flow1 = predict_flow(32)
        self.predict_flow0 = predict_flow(16)
        self.upsampled_flow2_to_1 = ConvTranspose2dF32(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = ConvTranspose2dF32(2, 2, 4, 2, 1)

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        flow2 = self.predict_flow2(c2)
        cat1 = torch.cat((c1, self.deconv1(c2), self.upsampled_flow2_to_1(flow2)), 1)
        flow1 = self.predict_flow1(self.inter_conv1(cat1))
        cat0 = torch.cat((c0, self.deconv0(cat1), self.upsampled_flow1_to_0(flow1)), 1)
        return self.predict_flow0(self.inter_conv0(cat0))


class FlowNet2(nn.Module):
    """models.py:25-128.  forward(inputs [B,3,2,H,W] in 0..255) -> flow [B,2,H,W]."""

    def __init__(self, batchNorm=False, div_flow=20.0):
        super().__init__()
        if batchNorm:
            raise NotImplementedError("the path builds FlowNet2 without batch norm (models.py:28)")
        self.div_flow = div_flow
        self.rgb_max = 255.0
        self.channelnorm = ops.ChannelNorm()
        self.flownetc = FlowNetC()
        self.resample1 = ops.Resample2d()
        self.flownets_1 = FlowNetS()
        self.resample2 = ops.Resample2d()
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.resample3 = ops.Resample2d()
        self.resample4 = ops.Resample2d()
        self.flownetfusion = FlowNetFusion()

    def forward(self, inputs):
        # Normalisation, flows and the warp/norm kernels stay float32 (as in the reference); only the convolution
        # trunks follow the module's parameter dtype (float32, or float16 on the throughput configuration).
        dt = self.flownetfusion.predict_flow0.weight.dtype
        inputs = inputs.float()
        mean = inputs.contiguous().view(inputs.shape[:2] + (-1,)).mean(dim=-1).view(inputs.shape[:2] + (1, 1, 1))
        x = (inputs - mean) / self.rgb_max
        x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1).contiguous()
        xt = x.to(dt)
        up_bil = lambda t: F.interpolate(t.float(), scale_factor=4, mode="bilinear")
        up_nn = lambda t: F.interpolate(t.float(), scale_factor=4, mode="nearest")

        flow_c = up_bil(self.flownetc(xt) * self.div_flow)
        concat1 = ops.warp_concat(x, flow_c, self.div_flow)          # models.py:86-91 in one kernel
        flow_s1 = up_bil(self.flownets_1(concat1.to(dt)) * self.div_flow)
        concat2 = ops.warp_concat(x, flow_s1, self.div_flow)         # :98-103
        flow_s2 = up_nn(self.flownets_2(concat2.to(dt)) * self.div_flow)
        n_s2, d_s2 = ops.warp_norms(x, flow_s2)                      # :108-112
        flow_sd = up_nn(self.flownets_d(xt) / self.div_flow)         # :115-116 (divided)
        n_sd, d_sd = ops.warp_norms(x, flow_sd)                      # :117-121
        concat3 = torch.cat((x[:, :3], flow_sd, flow_s2, n_sd, n_s2, d_sd, d_s2), dim=1)
        return self.flownetfusion(concat3.to(dt)).float()


class FlowProjectionModule(nn.Module):
    """FlowProjectionModule.py:9-33: two [h,w,3] frames -> Middlebury picture [h',w',3] (h',w' = multiples of 64)."""

    def __init__(self, image_size=None, render_size=None):
        super().__init__()
        self.net = FlowNet2()
        self.image_size = image_size
        self.render_size = render_size

    def _crop_pair(self, input1, input2):
        h, w = input1.shape[:2]
        th, tw = (h // 64) * 64, (w // 64) * 64
        if th == 0 or tw == 0:
            raise ValueError("FlowNet2 needs frames of at least 64x64 (centre crop to multiples of 64)")
        self.image_size, self.render_size = (h, w), [th, tw]
        y0, x0 = (h - th) // 2, (w - tw) // 2  # StaticCenterCrop, utils/tools.py:8-14
        images = torch.stack([input1[y0:y0 + th, x0:x0 + tw], input2[y0:y0 + th, x0:x0 + tw]])  # [2,h',w',3]
        return images.permute(3, 0, 1, 2)  # [3,2,h',w']

    @torch.no_grad()
    def flow(self, input1, input2, net=None):
        return (net or self.net)(self._crop_pair(input1, input2).unsqueeze(0))[0]  # [2,h',w']

    @torch.no_grad()
    def forward_pairs(self, pairs, net=None):
        """Several frame pairs as ONE FlowNet2 batch (independent samples: same arithmetic per pair, fewer and
        better filled launches) -> colour pictures [B,h',w',3].  `net`: an execution copy of self.net (e.g. fp16)."""
        if net is not None and hasattr(net, "run_pairs"):
            # fp16 executor: the frames go in as they are (crop, normalisation and layout inside prepare_pairs), the flow
            # comes back as the fusion network's NHWC half map and is colour-coded from there
            distinct, keys = [], []
            for t in (t for pr in pairs for t in pr):
                k = (t.data_ptr(), tuple(t.shape))
                if k not in keys:
                    keys.append(k)
                    distinct.append(t)
            h, w = distinct[0].shape[:2]
            th, tw = (h // 64) * 64, (w // 64) * 64
            if th == 0 or tw == 0:
                raise ValueError("FlowNet2 needs frames of at least 64x64 (centre crop to multiples of 64)")
            self.image_size, self.render_size = (h, w), [th, tw]
            idx = [(keys.index((a.data_ptr(), tuple(a.shape))), keys.index((b.data_ptr(), tuple(b.shape)))) for a, b in pairs]
            out = net.run_pairs(torch.stack(distinct), idx, ((h - th) // 2, (w - tw) // 2, th, tw))   # StaticCenterCrop, tools.py:8-14
            pics = torch.empty((len(pairs), th, tw, 3), dtype=torch.float32, device=out.device)
            for b in range(len(pairs)):
                ops.flow2img_nhwc(out[b], out=pics[b])   # the colour coding normalises by a per-picture maximum
            return pics
        batch = torch.stack([self._crop_pair(a, b) for a, b in pairs])  # [B,3,2,h',w']
        flows = (net or self.net)(batch)
        return torch.stack([ops.flow2img(f) for f in flows])  # the colour coding normalises by a per-picture maximum

    @torch.no_grad()
    def forward(self, input1, input2):
        return ops.flow2img(self.flow(input1, input2))
