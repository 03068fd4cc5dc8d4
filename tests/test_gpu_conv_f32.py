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


@pytest.mark.parametrize("case", [  # (N, C, H, W, Co, kh, kw, pad_y, pad_x, ctot, coff, bn, slope)
    (1, 64, 30, 50, 16, 11, 11, 5, 5, 16, 0, False, None),   # thin 11x11 (v_mfma_f32_16x16x4_f32)
    (2, 32, 21, 70, 16, 7, 7, 3, 3, 48, 32, True, 0.0),      # thin, into a concat slice, BatchNorm + ReLU
    (1, 64, 17, 33, 1, 3, 3, 1, 1, 1, 0, False, None),       # one out-channel
    (1, 32, 20, 40, 32, 7, 7, 3, 3, 96, 32, True, 0.0),      # 32 out-channels: 16-row tiles
    (2, 64, 19, 37, 64, 3, 3, 1, 1, 64, 0, False, 0.1),      # 64 out-channels, LeakyReLU
    (1, 128, 18, 64, 128, 3, 3, 1, 1, 128, 0, False, 0.0),   # 128 out-channels, 8-channel chunks
    (1, 64, 12, 40, 128, 5, 5, 2, 2, 128, 0, False, None),   # 128 out-channels, 4-channel chunks (weight block of a kernel row)
    (1, 6, 10, 35, 64, 3, 3, 1, 1, 64, 0, False, None),      # 6 input channels (padded to 16: the zero chunks are skipped)
    (1, 473, 8, 16, 256, 3, 3, 1, 1, 256, 0, False, 0.1),    # odd channel count, two 128-channel blocks
    (1, 64, 16, 32, 64, 11, 11, 5, 5, 64, 0, True, 0.0),     # 11x11 x 64 out-channels: 4-channel chunks
    (1, 24, 9, 31, 40, 3, 5, 1, 2, 40, 0, False, None),      # rectangular kernel, 40 out-channels (padded to 64)
    (1, 32, 7, 20, 24, 3, 3, 0, 0, 24, 0, False, None),      # no padding: the output is smaller than the input
])
def test_conv2d_f32_spatial_kernels_match_float64(case):
    """The spatial-reuse kernels (vsr_conv2d_act_nchw_f32, route 2) with the folded BatchNorm / activation / concat-slice epilogue
    against Conv2d -> BatchNorm2d(eval) -> (Leaky)ReLU in float64; the flat kernel (route 1) through the same entry too."""
    N, C, H, W, Co, kh, kw, py, px, ctot, coff, bn, slope = case
    rs = np.random.RandomState(C + 7 * Co + kh)
    x = torch.from_numpy(rs.randn(N, C, H, W).astype(np.float32))
    w = torch.from_numpy((rs.randn(Co, C, kh, kw) / np.sqrt(C * kh * kw)).astype(np.float32))
    b = torch.from_numpy(rs.randn(Co).astype(np.float32))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=(py, px))
    scale = shift = None
    if bn:
        mean, var = torch.from_numpy(rs.randn(Co)).double() * 0.3, torch.from_numpy(rs.rand(Co) + 0.5).double()
        gamma, beta = torch.from_numpy(rs.rand(Co) + 0.5).double(), torch.from_numpy(rs.randn(Co)).double()
        ref = F.batch_norm(ref, mean, var, gamma, beta, False, 0.0, 1e-5)
        sc = gamma / torch.sqrt(var + 1e-5)
        scale, shift = sc.float().cuda(), (beta + (b.double() - mean) * sc).float().cuda()
    else:
        shift = b.cuda()
    if slope is not None:
        ref = F.leaky_relu(ref, slope)
    wp = trunk_f32._pack(w.cuda().contiguous())
    for route in (2, 1):
        out = torch.full((N, ctot, ref.shape[2], ref.shape[3]), 7.0, dtype=torch.float32, device="cuda")
        trunk_f32.conv2d_fused(x.cuda(), wp, scale, shift, slope is not None, slope or 0.0, Co, kh, kw, 1, py, px, route, out=out, coff=coff)
        got = out[:, coff:coff + Co].cpu().double()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err <= TOL, (route, err)
        rest = torch.cat([out[:, :coff], out[:, coff + Co:]], 1)
        assert (rest == 7.0).all(), "channels outside the slice were written"


def test_thin_layer_with_a_kernel_wider_than_16_leaves_the_thin_spatial_kernel(monkeypatch):
    """k_conv_f32_sp16 stages one kernel row of kw x 4 x 16 weights with one float4 per thread (256 threads:
    kw <= 16).  A <= 16-out-channel layer with kw = 17 must not be routed there (router and launcher), and is still served correctly."""
    monkeypatch.setattr(trunk_f32, "MIN_TILES", 0)
    monkeypatch.setattr(trunk_f32, "MIN_WGS", 0)
    N, C, H, W, Co, k = 1, 32, 24, 70, 16, 17
    assert trunk_f32._route(N, C, H, W, Co, k, k, 1, 8, 8) != trunk_f32.SPATIAL_K
    assert trunk_f32._route(N, C, H, W, Co, 11, 11, 1, 5, 5) == trunk_f32.SPATIAL_K
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(N, C, H, W).astype(np.float32))
    w = torch.from_numpy((rs.randn(Co, C, k, k) / np.sqrt(C * k * k)).astype(np.float32))
    b = torch.from_numpy(rs.randn(Co).astype(np.float32))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=8)
    wp = trunk_f32._pack(w.cuda().contiguous())
    from video_super_resolution_amd import _lib as L
    with pytest.raises(L.VsrHipError):   # the launcher refuses the spatial route for it (it used to compute with uninitialised LDS)
        trunk_f32.conv2d_fused(x.cuda(), wp, None, b.cuda(), False, 0.0, Co, k, k, 1, 8, 8, 2)
    for route in (0, 1):
        got = trunk_f32.conv2d_fused(x.cuda(), wp, None, b.cuda(), False, 0.0, Co, k, k, 1, 8, 8, route).cpu().double()
        assert (got - ref).abs().max().item() <= TOL * ref.abs().max().item(), route
    m = trunk_f32.Conv2dF32(C, Co, k, 1, 8).cuda()
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
        got = m(x.cuda()).cpu().double()
    assert (got - ref).abs().max().item() <= TOL * ref.abs().max().item()


def test_fused_sequential_and_concat_match_the_separate_passes(monkeypatch):
    """FusedSequential / depth.ChannelConcat (Conv2d -> BatchNorm2d -> ReLU in one launch, branches written into the concat buffer in
    place) against the same modules evaluated child by child on the stock operators."""
    from video_super_resolution_amd import depth
    monkeypatch.setattr(trunk_f32, "MIN_TILES", 0)
    monkeypatch.setattr(trunk_f32, "MIN_WGS", 0)
    torch.manual_seed(3)
    blk = depth._build(depth._J).cuda().eval()    # 128 -> 16 + 3 x (64 -> 16, k 3 / 7 / 11)
    seq = depth._build(("S", [("conv", 3, 128, 7, 3), ("bn", 128, True), "relu", depth._J, ("conv", 64, 1, 3, 1)])).cuda().eval()
    with torch.no_grad():
        for m in list(blk.modules()) + list(seq.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
        x = torch.randn(2, 128, 37, 45, device="cuda")
        img = torch.randn(1, 3, 40, 70, device="cuda")
        got, got2 = blk(x), seq(img)
        monkeypatch.setattr(trunk_f32, "FUSE", False)
        monkeypatch.setattr(trunk_f32, "ENABLED", False)
        ref, ref2 = blk(x), seq(img)
    assert got.shape == ref.shape == (2, 64, 37, 45) and got2.shape == ref2.shape
    assert (got - ref).abs().max().item() <= TOL * ref.abs().max().item()
    assert (got2 - ref2).abs().max().item() <= 5 * TOL * ref2.abs().max().item()


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
