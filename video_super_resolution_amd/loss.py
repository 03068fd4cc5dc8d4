"""The `train=True` branch of `VSR.forward`: the reference's loss modules behind their own names.

Mirrors loss_function.py:9-101 and `VSR.loss_calculate` (network/video_super_resolution.py:71-80):

    SR_loss               MSE + 0.006 * MSE(VGG16 features[:31]) + 2e-8 * TV                     (loss_function.py:9-28, :31-48)
    Flow_loss             0.005 * mean(SR_loss(f0, f1), SR_loss(f1, f2)) with its OWN SR_loss       (:51-62)
    GetObjectsForOBJLoss  OSVOS mask of (frame 0, frame 1), computed ONCE and cached forever (D7, :69-74), applied by
                          numpy masked arrays whose [3,H,W] mask is re-read as [H,W,3] (equal sizes: numpy reshapes it),
                          on frames cast to uint8 (C cast: truncation, wrap modulo 256), masked entries filled with 0 in both
                          variants (`fill_value=0`: getSRMaskedOutputs :89,:92 and getFlowMaskedOutputs :99)
    loss = genSR + objSR + 0.006 * genFlow + 0.006 * objFlow                                       (:73-80), a 0-d CPU tensor

Sub-module and parameter names are the reference's (`SR_loss.loss_network.<i>`, `Flow_loss.SR_loss.loss_network.<i>`,
`loss4object.VOS.net.*`), so a full-model `state_dict` interchanges.  Everything runs on the module's device under
`no_grad` (the reference computes the loss under no_grad too, :72).  Like the guidance trunks, the VGG16 feature networks and
the loss's own OSVOS follow `VSR.precision`: "fp16" = the hand-written NHWC fp16 MFMA convolution (trunk_exec.VGGFeatExec /
OSVOSExec), "fp32" = stock float32 convolutions.  The loss is not on the inference path the benchmark times; it exists so that
the reference driver's own call `model(x, y, high_frame, estimated_image)` (main.py:201, train defaults to True) works and
returns the same number.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .vos import VOSProjectionModule

_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def vgg16_features31() -> nn.Sequential:
    """`list(vgg16().features)[:31]` (utils/models.py:60-79, loss_function.py:13): 13 conv3x3 + ReLU, 5 MaxPool2d(2, 2);
    module indices as in torchvision's VGG16 so the parameter names match."""
    layers, cin = [], 3
    for v in _VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    seq = nn.Sequential(*layers[:31])
    for p in seq.parameters():
        p.requires_grad = False
    return seq


class TVLoss(nn.Module):
    def __init__(self, tv_loss_weight=1):
        super().__init__()
        self.tv_loss_weight = tv_loss_weight

    def forward(self, x):
        b, c, h, w = x.shape
        count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
        h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum()
        w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum()
        return self.tv_loss_weight * 2 * (h_tv / count_h + w_tv / count_w) / b


class SR_loss(nn.Module):   # noqa: N801 (the reference's class name)
    def __init__(self):
        super().__init__()
        self.loss_network = vgg16_features31()
        self.mse_loss = nn.MSELoss()
        self.tv_loss = TVLoss()
        # "fp16": the VGG16 trunk on the hand-written NHWC fp16 MFMA convolution (trunk_exec.VGGFeatExec), like the guidance
        # trunks of the fp16 configuration; "fp32": stock float32 convolutions.  Set by VSR.forward from VSR.precision.
        self.precision = "fp32"
        self._exec = None

    def _features(self, x):
        if self.precision == "fp16" and x.is_cuda:
            from .trunk_exec import TrunkExecCache, VGGFeatExec
            if self._exec is None:
                self._exec = TrunkExecCache(self.loss_network, VGGFeatExec)
            return self._exec.get()(x.contiguous())
        return self.loss_network(x)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_exec" else copy.deepcopy(v, memo)   # (executors hold per-instance buffers)
        return new

    def forward(self, output, target):
        dev = next(self.loss_network.parameters()).device
        output = output.to(dev, torch.float32).permute(0, 3, 1, 2)   # transpose1323
        target = target.to(dev, torch.float32).permute(0, 3, 1, 2)
        perception_loss = self.mse_loss(self._features(output), self._features(target))
        image_loss = self.mse_loss(output, target)
        return image_loss + 0.006 * perception_loss + 2e-8 * self.tv_loss(output)


class Flow_loss(nn.Module):   # noqa: N801
    def __init__(self):
        super().__init__()
        self.mse_loss = nn.MSELoss()
        self.SR_loss = SR_loss()

    def forward(self, outputs):
        flow_loss = torch.mean(torch.stack((self.SR_loss(outputs[0:1], outputs[1:2]), self.SR_loss(outputs[1:2], outputs[2:3]))))
        return 0.005 * flow_loss


def _as_uint8(x: torch.Tensor) -> torch.Tensor:
    """`np.array(t.cpu(), dtype=np.uint8)` on float32 data: the C cast numpy performs -- truncation toward zero, then the
    low eight bits (255.9 -> 255, 256.5 -> 0, 300.2 -> 44; measured with numpy 2.2 on x86-64)."""
    return (x.to(torch.float32).to(torch.int64) & 255).to(torch.float32)


class GetObjectsForOBJLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.VOS = VOSProjectionModule().eval()
        self.mask = None   # [3,H,W] bool, cached after the first call like the reference (loss_function.py:69-74, defect D7)
        self.precision = "fp32"   # "fp16": OSVOS on the MFMA convolution (trunk_exec.OSVOSExec), as in the inference path
        self._exec = None

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_exec" else copy.deepcopy(v, memo)
        return new

    def reset(self):
        self.mask = None

    def forward(self, outputs, target=None, SR=False):   # noqa: N803
        if self.mask is None:
            net = None
            if self.precision == "fp16" and outputs[0].is_cuda:
                from .trunk_exec import OSVOSExec, TrunkExecCache
                if self._exec is None:
                    self._exec = TrunkExecCache(self.VOS.net, OSVOSExec)
                net = self._exec.get()
            seg = self.VOS(outputs[0].to(torch.float32), outputs[1].to(torch.float32), net)   # [H,W] in {0,1}
            self.mask = torch.stack((seg == 1,) * 3)                                    # maskprocess(obj_segmentation == 1)
        if SR:
            # np.ma.MaskedArray(data [H,W,3] | [1,H,W,3], mask [3,H,W]): equal sizes -> numpy RESHAPES the mask to the data's shape
            m_out = self.mask.reshape(outputs[1].shape)
            masked_output = torch.where(m_out, torch.zeros((), device=m_out.device), _as_uint8(outputs[1])).unsqueeze(0)
            m_tgt = self.mask.reshape(target.shape)
            masked_target = torch.where(m_tgt, torch.zeros((), device=m_tgt.device), _as_uint8(target))
            return masked_output, masked_target
        # getFlowMaskedOutputs (loss_function.py:95-101): `fill_value=0` like the SR variant
        outs = [torch.where(self.mask.reshape(o.shape), torch.zeros((), device=o.device), _as_uint8(o)) for o in outputs]
        return torch.stack(outs).to(torch.float32)


def loss_calculate(model, target, outputs, taps: dict | None = None) -> torch.Tensor:
    """VSR.loss_calculate (network/video_super_resolution.py:71-80).  target [1,H,W,3], outputs = high_frames [3,H,W,3].
    `taps`, when given, receives the four terms (genSR, objSR, genFlow, objFlow) and the tensors loss4object returned, so that
    tests can pin each against the reference's (fixture g10) rather than only their weighted sum."""
    with torch.no_grad():
        model.SR_loss.precision = model.Flow_loss.SR_loss.precision = model.loss4object.precision = getattr(model, "precision", "fp32")
        gen_sr = model.SR_loss(outputs[0:1], target)
        masked = model.loss4object(outputs[:2], target, SR=True)
        obj_sr = model.SR_loss(masked[0], masked[1])
        gen_flow = model.Flow_loss(outputs)
        masked_flow = model.loss4object(outputs)
        obj_flow = model.Flow_loss(masked_flow)
        if taps is not None:
            taps.update(terms=[float(gen_sr), float(obj_sr), float(gen_flow), float(obj_flow)], masked_sr_out=masked[0],
                        masked_sr_tgt=masked[1], masked_flow=masked_flow)
        loss = gen_sr.data + obj_sr.data + 0.006 * gen_flow.data + 0.006 * obj_flow.data
        return loss.cpu()
