"""The float32 NCHW convolution on the matrix cores (csrc/conv_f32_nchw.hip, `trunk_f32.Conv2dF32` / `ConvTranspose2dF32`: what the
guidance trunks' nn.Conv2d / nn.ConvTranspose2d layers run on in the float32 configuration) against the same operator evaluated in
float64 on the CPU: float32 products summed in float32 in another order than any reference -> bar 2e-5 of the output range (the
float32 bar of the SR kernels).  Shapes: every kind of layer the three trunks hold."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from video_super_resolution_amd import trunk_f32  # noqa: E402

TOL = 2e-5


@pytest.mark.parametrize("case", [  # (N, C, H, W, Co, k, stride, pad, bias)
    (1, 3, 37, 45, 128, 7, 1, 3, True),      # hourglass stem
    (2, 6, 40, 52, 64, 7, 2, 3, True),       # FlowNetC conv1 on a pair
    (2, 128, 19, 23, 32, 1, 1, 0, True),     # inception 1x1
    (1, 64, 30, 50, 16, 11, 1, 5, True),     # 16-wide 11x11 branch
    (1, 473, 8, 16, 256, 3, 1, 1, True),     # FlowNetC conv3_1 (odd channel count)
    (2, 1026, 4, 6, 2, 3, 1, 1, True),       # predict_flow
    (1, 256, 33, 47, 512, 3, 2, 1, True),    # stride 2, 4 blocks of 128 out-channels
    (1, 32, 5, 300, 48, 3, 1, 1, False),     # rows wider than one 128-column segment, no bias, 48 -> one 64-channel block
    (3, 64, 6, 5, 1, 3, 1, 1, True),         # one out-channel
    (1, 16, 9, 140, 16, 1, 1, 0, True),      # OSVOS score layer shape
])
def test_conv2d_f32_matches_float64(case, monkeypatch):
    monkeypatch.setattr(trunk_f32, "ROUTE", False)     # every shape through the own kernel (the router would keep the small ones on the stock operator)
    N, C, H, W, Co, k, stride, pad, bias = case
    rs = np.random.RandomState(C + 7 * Co + k)
    x = torch.from_numpy(rs.randn(N, C, H, W).astype(np.float32))
    m = trunk_f32.Conv2dF32(C, Co, k, stride, pad, bias=bias)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy((rs.randn(Co, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)))
        if bias:
            m.bias.copy_(torch.from_numpy(rs.randn(Co).astype(np.float32)))
        ref = F.conv2d(x.double(), m.weight.double(), m.bias.double() if bias else None, stride=stride, padding=pad)
        cpu = m(x)                                   # CPU tensor: the stock operator
        assert cpu.dtype == torch.float32 and (cpu.double() - ref).abs().max() <= 1e-4 * ref.abs().max()
        m = m.cuda()
        got = m(x.cuda())
        assert got.shape == ref.shape
        err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= TOL, err
        # a changed parameter is packed again
        m.weight.mul_(2.0)
        got2 = m(x.cuda())
        ref2 = F.conv2d(x.double(), m.weight.cpu().double(), m.bias.cpu().double() if bias else None, stride=stride, padding=pad)
        assert (got2.cpu().double() - ref2).abs().max().item() <= TOL * ref2.abs().max().item()
    # under autograd the stock operator serves (gradients exist)
    xg = x.cuda().requires_grad_()
    y = m(xg)
    y.sum().backward()
    assert xg.grad is not None and xg.grad.shape == xg.shape


@pytest.mark.parametrize("case", [(2, 64, 7, 9, 32, True), (1, 1026, 4, 5, 256, True), (1, 2, 16, 30, 2, False), (2, 386, 16, 30, 64, True)])
def test_conv_transpose_k4s2_f32_matches_float64(case, monkeypatch):
    monkeypatch.setattr(trunk_f32, "ROUTE", False)
    N, C, H, W, Co, bias = case
    rs = np.random.RandomState(C + Co)
    x = torch.from_numpy(rs.randn(N, C, H, W).astype(np.float32))
    m = trunk_f32.ConvTranspose2dF32(C, Co, 4, 2, 1, bias=bias)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy((rs.randn(C, Co, 4, 4) / np.sqrt(4 * C)).astype(np.float32)))
        if bias:
            m.bias.copy_(torch.from_numpy(rs.randn(Co).astype(np.float32)))
        ref = F.conv_transpose2d(x.double(), m.weight.double(), m.bias.double() if bias else None, stride=2, padding=1)
        m = m.cuda()
        got = m(x.cuda())
    assert got.shape == ref.shape == (N, Co, 2 * H, 2 * W)
    err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= TOL, err


def test_trunks_in_float32_run_on_the_own_kernel(gpu_vsr):
    """The three master trunks on a CUDA float32 input: every nn.Conv2d (and FlowNet's k4 s2 transposed convolutions) is served by
    csrc/conv_f32_nchw.hip, and the results agree with the stock operators (trunk_f32.ENABLED = False) at the float32 bar."""
    rs = np.random.RandomState(3)
    fr = torch.from_numpy(rs.randint(0, 256, (2, 3, 64, 128)).astype(np.float32)).cuda()
    nets = {"depth": (gpu_vsr.DepthModule.model.netG, fr), "vos": (gpu_vsr.VOSModule.net, fr - 110.0),
            "flow": (gpu_vsr.FlowModule.net, fr.view(1, 2, 3, 64, 128).permute(0, 2, 1, 3, 4).contiguous())}
    for name, (net, x) in nets.items():
        with torch.no_grad():
            own = net(x)
            trunk_f32.ENABLED = False
            try:
                stock = net(x)
            finally:
                trunk_f32.ENABLED = True
        own, stock = (own[0] if isinstance(own, (tuple, list)) else own), (stock[0] if isinstance(stock, (tuple, list)) else stock)
        err = (own - stock).abs().max().item() / stock.abs().max().item()
        print(f"[{name} fp32 trunk, own kernel vs stock operators] max err / range = {err:.3e}")
        assert err <= 1e-3, (name, err)
