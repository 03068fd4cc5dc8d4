"""The generic NHWC fp16 MFMA convolution (csrc/conv_igemm.hip) against torch.nn.functional on the same fp16-rounded
operands (fp32 accumulate on both sides): bar 2e-3 of the output range (fp16 output rounding is 4.9e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from video_super_resolution_amd import igemm  # noqa: E402

CASES = [  # (N, Cin, H, W, Cout, k, stride, pad, act)
    (1, 3, 20, 28, 128, 7, 1, 3, igemm.ACT_RELU),      # hourglass stem
    (2, 128, 17, 23, 32, 1, 1, 0, igemm.ACT_RELU),     # inception 1x1
    (1, 32, 19, 21, 32, 7, 1, 3, igemm.ACT_RELU),      # inception 7x7
    (1, 64, 9, 11, 64, 11, 1, 5, igemm.ACT_RELU),      # inception 11x11
    (2, 12, 32, 40, 64, 7, 2, 3, igemm.ACT_LEAKY),     # FlowNetS conv1
    (1, 473, 8, 16, 256, 3, 1, 1, igemm.ACT_LEAKY),    # FlowNetC conv3_1 (odd channel count)
    (1, 1026, 4, 6, 2, 3, 1, 1, igemm.ACT_NONE),       # predict_flow (2 outputs)
    (3, 64, 6, 5, 1, 3, 1, 1, igemm.ACT_NONE),         # single output channel
    (1, 256, 33, 47, 512, 3, 2, 1, igemm.ACT_LEAKY),   # stride 2
    (1, 96, 130, 3, 48, 3, 1, 1, igemm.ACT_RELU),      # narrow, more pixels than one tile
    (1, 64, 92, 94, 256, 3, 2, 1, igemm.ACT_LEAKY),    # stride 2, 256 out-channels (128-channel tiles when forced below)
    (2, 128, 131, 67, 128, 5, 2, 2, igemm.ACT_RELU),   # 5x5 stride 2, ragged
    (4, 64, 128, 130, 128, 3, 2, 1, igemm.ACT_LEAKY),  # enough workgroups for the 128-channel gather tile by default
    (2, 64, 37, 61, 1, 3, 1, 1, igemm.ACT_NONE),       # hourglass final conv: LDS-patch path (cout <= 16, stride 1)
    (1, 64, 20, 50, 16, 11, 1, 5, igemm.ACT_RELU),     # 16-wide 11x11 inception branch: patch path
    (1, 194, 33, 47, 2, 3, 1, 1, igemm.ACT_NONE),      # predict_flow2: patch path over 7 channel chunks
    (1, 32, 9, 40, 16, 7, 1, 3, igemm.ACT_RELU),       # patch path, ragged tile edges
    (1, 128, 259, 271, 208, 1, 1, 0, igemm.ACT_RELU),  # fused inception 1x1s, many pixels: streaming 1x1 kernel (ragged last block)
    (2, 256, 190, 181, 160, 1, 1, 0, igemm.ACT_RELU),  # streaming 1x1, 8 channel chunks
    (1, 96, 300, 230, 77, 1, 1, 0, igemm.ACT_LEAKY),   # streaming 1x1, odd out-channel count
]


@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch(case):
    N, cin, H, W, cout, k, s, p, act = case
    rs = np.random.RandomState(cin * 7 + cout)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=s, padding=p)
    if act == igemm.ACT_RELU:
        ref = F.relu(ref)
    elif act == igemm.ACT_LEAKY:
        ref = F.leaky_relu(ref, 0.1)
    from video_super_resolution_amd import _lib as L
    conv = igemm.HConv(w, b, stride=s, pad=p, act=act)
    out = conv(igemm.to_nhwc_half(x))
    route = L.load().vsr_last_route().decode()
    got = igemm.to_nchw_float(out, cout)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item(), (err, ref.abs().max().item())
    if out.shape[3] > cout:  # padding channels stay zero so the tensor can feed the next layer
        assert float(out[..., cout:].abs().max()) == 0.0
    # the first gather build (pixel operand through LDS, 64-channel tiles) runs the same MFMAs in the same order: equal bit for bit
    old = L.load().vsr_conv2d_tuning(8)
    try:
        out8 = conv(igemm.to_nhwc_half(x))
        torch.cuda.synchronize()
    finally:
        L.load().vsr_conv2d_tuning(old)
    if route.startswith("gather") and (cout <= 64 or (N * out.shape[1] * out.shape[2] + 127) // 128 * ((cout + 127) // 128) < 128):
        assert torch.equal(out8, out)   # (128-channel tiles may split K differently: a different fp32 summation order)
    else:
        assert (igemm.to_nchw_float(out8, cout) - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    for mode in (10, 11):   # 128-channel gather tiles wherever the channel count allows / nowhere
        old = L.load().vsr_conv2d_tuning(mode)
        try:
            outm = conv(igemm.to_nhwc_half(x))
            torch.cuda.synchronize()
        finally:
            L.load().vsr_conv2d_tuning(old)
        assert (igemm.to_nchw_float(outm, cout) - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), mode


TILE_CASES = [  # (N, Cin, H, W, Cout, k, stride, pad, act): the two-operand LDS-DMA tile kernel (csrc/conv_tile.hip)
    (1, 256, 33, 47, 512, 3, 2, 1, igemm.ACT_LEAKY),   # stride 2, 128-channel tiles, ragged pixel tile, split-K
    (2, 512, 34, 60, 512, 3, 1, 1, igemm.ACT_RELU),    # OSVOS's VGG 512 -> 512 at 34 x 60
    (1, 473, 8, 16, 256, 3, 1, 1, igemm.ACT_LEAKY),    # odd channel count (15 chunks: the odd tail pair is zero-filled), one pixel tile
    (2, 128, 131, 67, 128, 5, 2, 2, igemm.ACT_RELU),   # 5x5 stride 2
    (1, 64, 92, 94, 256, 3, 2, 1, igemm.ACT_LEAKY),    # 64 -> 256
    (4, 64, 128, 130, 128, 3, 2, 1, igemm.ACT_LEAKY),  # many workgroups (the XCD remap's padded grid)
    (1, 96, 40, 52, 64, 3, 1, 1, igemm.ACT_NONE),      # 64-channel tile, three chunks
    (1, 32, 67, 45, 70, 7, 1, 3, igemm.ACT_RELU),      # cout 70 -> padded 128: dead out-channel rows
    (3, 160, 9, 7, 192, 1, 1, 0, igemm.ACT_RELU),      # 1x1, 192 out-channels (64-channel tiles), images smaller than a tile
    (1, 1056, 16, 30, 512, 3, 1, 1, igemm.ACT_LEAKY),  # long K (297 pairs), split-K
]


@pytest.mark.parametrize("case", TILE_CASES)
@pytest.mark.parametrize("mode", [2003])
def test_tile_kernel_matches_torch(case, mode):
    """csrc/conv_tile.hip against the stock operator on the same fp16-rounded operands, and -- with split-K off on both sides --
    against the gather kernel k_conv_igemm_d bit for bit: same products, same K order per accumulator (fp32 accumulate in
    MFMA 16x16x32 steps of one (tap, 32-channel chunk) pair each)."""
    from video_super_resolution_amd import _lib as L
    N, cin, H, W, cout, k, s, p, act = case
    rs = np.random.RandomState(cin * 3 + cout + k)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=s, padding=p)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=s, pad=p, act=act)
    lib = L.load()
    xs = igemm.to_nhwc_half(x)
    lib.vsr_conv2d_tuning(mode)
    try:
        out = conv(xs).clone()
        route = lib.vsr_last_route().decode()
        assert route.startswith("tile<"), route
        got = igemm.to_nchw_float(out, cout)
        err = (got - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item(), (route, err, ref.abs().max().item())
        if out.shape[3] > cout:
            assert float(out[..., cout:].abs().max()) == 0.0
        assert torch.equal(conv(xs), out)                       # deterministic (split-K sums in a fixed order)
        # no split-K on either side: the tile kernel against the gather kernel, bit for bit
        lib.vsr_conv2d_tuning(5001)
        lib.vsr_conv2d_tuning(1000)
        t_ns = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("tile<") and "splitk" not in lib.vsr_last_route().decode()
        lib.vsr_conv2d_tuning(2000)
        lib.vsr_conv2d_tuning(1)                                # (mode 1: never the patch kernels)
        g_ns = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("gather<"), lib.vsr_last_route().decode()
        assert torch.equal(t_ns, g_ns)
    finally:
        lib.vsr_conv2d_tuning(0)
        lib.vsr_conv2d_tuning(2001)
        lib.vsr_conv2d_tuning(5000)
        lib.vsr_conv2d_tuning(1128)


@pytest.mark.parametrize("case", [  # (N, Cin, H, W, Cout, k, act): the LDS-patch builds, forced (the heuristic wants >= 8192 pixels)
    (1, 64, 37, 61, 16, 11, igemm.ACT_RELU), (2, 64, 20, 50, 16, 7, igemm.ACT_RELU), (1, 32, 9, 40, 16, 5, igemm.ACT_RELU),
    (1, 96, 33, 47, 16, 3, igemm.ACT_LEAKY), (2, 64, 37, 61, 1, 3, igemm.ACT_NONE), (1, 194, 33, 47, 2, 3, igemm.ACT_NONE),
    (1, 64, 41, 77, 32, 7, igemm.ACT_RELU), (1, 64, 19, 33, 64, 5, igemm.ACT_RELU), (1, 32, 130, 70, 13, 11, igemm.ACT_RELU),
    (2, 64, 33, 50, 128, 3, igemm.ACT_LEAKY), (1, 64, 30, 40, 64, 11, igemm.ACT_RELU), (1, 96, 12, 16, 48, 3, igemm.ACT_RELU)])
@pytest.mark.parametrize("mode", [2, 6, 7, 1])
def test_patch_kernels_match_torch(case, mode):
    """mode 2: every legal layer through the LDS-patch builds (k_conv_patch_r8 where one exists); 6: without r8
    (k_conv_patch_rows / k_conv_patch); 7: k_conv_patch only; 1: the same layers through the gather kernel."""
    from video_super_resolution_amd import _lib as L
    N, cin, H, W, cout, k, act = case
    rs = np.random.RandomState(cin + 31 * cout + k)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=1, padding=k // 2)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=1, pad=k // 2, act=act)
    old = L.load().vsr_conv2d_tuning(mode)
    try:
        out = conv(igemm.to_nhwc_half(x))
        torch.cuda.synchronize()
    finally:
        L.load().vsr_conv2d_tuning(old)
    got = igemm.to_nchw_float(out, cout)
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("case", [(2, 3, 37, 45, 128, 7, 1, 3, igemm.ACT_RELU), (4, 3, 64, 96, 64, 7, 2, 3, igemm.ACT_LEAKY),
                                  (1, 3, 33, 29, 64, 3, 1, 1, igemm.ACT_RELU), (1, 4, 20, 20, 16, 5, 1, 2, igemm.ACT_NONE),
                                  (1, 3, 270, 250, 128, 7, 1, 3, igemm.ACT_RELU),    # hourglass stem, >= 65536 pixels: k_stem7_rows (ragged tiles)
                                  (3, 3, 161, 144, 128, 7, 1, 3, igemm.ACT_LEAKY)])  # the same across image boundaries
def test_stem_conv_matches_torch(case):
    """Dense-K first convolution ([N,H,W,4] input, one K chunk per kernel row) of the hourglass / FlowNetC / VGG."""
    N, cin, H, W, cout, k, s, p, act = case
    rs = np.random.RandomState(cout + k)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=s, padding=p)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    out = igemm.HConvStem(w, b, stride=s, pad=p, act=act)(igemm.to_nhwc_half(x, 4))
    got = igemm.to_nchw_float(out, cout)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_conv_writes_into_channel_slices():
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(1, 64, 10, 12).astype(np.float32)).cuda().half()
    xs = igemm.to_nhwc_half(x)
    dst = torch.zeros((1, 10, 12, 96), dtype=torch.float16, device="cuda")
    convs = []
    for off, cout in ((0, 32), (32, 16), (48, 48)):
        w = torch.from_numpy((rs.randn(cout, 64, 3, 3) / 24).astype(np.float32)).cuda().half().float()
        c = igemm.HConv(w, None, stride=1, pad=1)
        c(xs, out=dst, out_coff=off)
        convs.append((off, cout, w))
    for off, cout, w in convs:
        ref = F.conv2d(x.float(), w, None, padding=1)
        got = dst[..., off:off + cout].permute(0, 3, 1, 2).float()
        assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(1, 64, 7, 9, 32), (2, 1026, 4, 5, 256), (1, 386, 16, 30, 64),
                                   # few out-channels on >= 8192 pixels: the four phases from one staged patch (k_deconv4s2_patch), ragged tiles
                                   (2, 192, 83, 101, 16), (1, 32, 97, 130, 2), (2, 128, 70, 118, 32), (1, 64, 128, 64, 13),
                                   # wide layers with >= 100 workgroups: the tile kernel, four phases in one launch (FlowNet decoders)
                                   (2, 416, 64, 120, 64), (2, 800, 32, 60, 128), (1, 96, 61, 47, 192)])
def test_transposed_conv_k4s2(shape):
    N, cin, H, W, cout = shape
    rs = np.random.RandomState(cin)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cin, cout, 4, 4) / np.sqrt(cin * 4)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.leaky_relu(F.conv_transpose2d(x.float(), w, b, stride=2, padding=1), 0.1)
    from video_super_resolution_amd import _lib as L
    dc = igemm.HDeconv4s2(w, b, act=igemm.ACT_LEAKY)
    got = igemm.to_nchw_float(dc(igemm.to_nhwc_half(x)), cout)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    from video_super_resolution_amd import _lib as L
    route = L.load().vsr_last_route().decode()
    lib = L.load()
    if cout > 32:    # the tile kernel on the same layer (forced: the heuristic takes only the long-K / many-workgroup ones), four phases in one launch
        lib.vsr_conv2d_tuning(2003)
        try:
            gott = igemm.to_nchw_float(dc(igemm.to_nhwc_half(x)), cout)
            assert "tile<" in lib.vsr_last_route().decode()
        finally:
            lib.vsr_conv2d_tuning(2001)
        assert (gott - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    old = L.load().vsr_conv2d_tuning(1)   # the same layer through the gather kernel (four phases in one launch)
    try:
        got1 = igemm.to_nchw_float(dc(igemm.to_nhwc_half(x)), cout)
        assert "gather<" in L.load().vsr_last_route().decode()
    finally:
        L.load().vsr_conv2d_tuning(old)
    assert (got1 - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 64, 120, 256), (1, 13, 37, 256), (1, 8, 15, 64)])
def test_flownetc_cost_volume_mfma(shape):
    """MFMA cost volume + LeakyReLU written into a concat-buffer slice, against the float32 Correlation kernel (itself
    pinned to the reference's CUDA kernel by tests/test_gpu_flow_ops.py) on the same fp16-rounded features."""
    import ctypes
    from video_super_resolution_amd import _lib as L, ops
    B, H, W, C = shape
    rs = np.random.RandomState(H * W)
    a = torch.from_numpy(rs.randn(B, H, W, C).astype(np.float32)).cuda().half()
    b = torch.from_numpy(rs.randn(B, H, W, C).astype(np.float32)).cuda().half()
    ref = F.leaky_relu(ops.correlation(a.permute(0, 3, 1, 2).float().contiguous(), b.permute(0, 3, 1, 2).float().contiguous(),
                                       20, 1, 20, 1, 2), 0.1)                       # [B,441,H,W]
    out = torch.full((B, H, W, 480), 7.0, dtype=torch.float16, device="cuda")
    L.check(L.load().vsr_flownetc_corr_nhwc_f16(L.dptr(a, torch.float16), L.dptr(b, torch.float16), L.dptr(out, torch.float16), 480, 32,
                                                B, H, W, C, L.stream()))
    got = out[..., 32:473].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    assert float((out[..., :32] - 7.0).abs().max()) == 0.0 and float((out[..., 473:] - 7.0).abs().max()) == 0.0   # slice only


@pytest.mark.parametrize("shape", [(2, 135, 68, 64), (1, 7, 9, 16), (1, 34, 60, 128)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pool2x2(shape, mode):
    """2x2 stride-2 pools on NHWC fp16: max, average (floor) and max with ceil_mode (OSVOS's VGG pools)."""
    N, H, W, C = shape
    x = torch.from_numpy(np.random.RandomState(H + W + mode).randn(N, H, W, C).astype(np.float32)).cuda().half()
    got = igemm.pool2x2(x, 0, C, mode)
    xn = x.permute(0, 3, 1, 2).float()
    ref = F.avg_pool2d(xn, 2, 2) if mode == 1 else F.max_pool2d(xn, 2, 2, ceil_mode=(mode == 2))
    ref = ref.permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    if mode == 1:
        assert (got.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    else:
        assert torch.equal(got.float(), ref)


@pytest.mark.parametrize("shape", [(2, 6, 37, 45, None), (1, 12, 64, 96, None), (2, 3, 19, 21, 4), (1, 11, 8, 8, 32), (1, 40, 5, 7, None)])
def test_nchw_f32_to_nhwc_f16(shape):
    """Trunk input conversion in one pass: same values as zero fill + strided copy (Tensor.half() rounding), padding zero."""
    N, C, H, W, cp = shape
    x = torch.from_numpy((np.random.RandomState(C + H).randn(N, C, H, W) * 100).astype(np.float32)).cuda()
    got = igemm.to_nhwc_half(x, cp)
    cpp = cp or igemm.pad32(C)
    ref = torch.zeros((N, H, W, cpp), dtype=torch.float16, device="cuda")
    ref[..., :C] = x.permute(0, 2, 3, 1)
    assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.parametrize("case", [(2, 12, 64, 96, 64, 7, 3), (1, 12, 37, 50, 64, 7, 3), (1, 16, 20, 34, 32, 5, 2), (1, 5, 9, 8, 16, 3, 1)])
def test_pair_conv_stride2(case):
    """Stride-2 first convolution of a <=16-channel map as a stride-(2,1) convolution over pixel pairs (FlowNetS conv1)."""
    N, cin, H, W, cout, k, pad = case
    rs = np.random.RandomState(cin + k + W)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.leaky_relu(F.conv2d(x.float(), w, b, stride=2, padding=pad), 0.1)
    conv = igemm.HConvPairS2(w, b, pad=pad, act=igemm.ACT_LEAKY, slope=0.1)
    out = conv(igemm.to_nhwc_half(x.float(), 16))
    got = igemm.to_nchw_float(out, cout)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_splitk_runs_are_bit_identical():
    """Split-K layers (few output pixels, long K): k_splitk_finish sums the partial tiles in the fixed order z = 0, 1, ..."""
    rs = np.random.RandomState(5)
    for N, cin, H, W, cout, k, s in ((1, 473, 8, 16, 256, 3, 1), (2, 512, 16, 30, 512, 3, 2), (1, 1026, 4, 6, 2, 3, 1)):
        x = igemm.to_nhwc_half(torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda())
        w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda()
        conv = igemm.HConv(w, torch.zeros(cout, device="cuda"), stride=s, pad=1, act=igemm.ACT_LEAKY)
        first = conv(x).clone()
        for _ in range(8):
            assert torch.equal(conv(x), first)
        ref = F.leaky_relu(F.conv2d(igemm.to_nchw_float(x, cin), w.half().float(), None, stride=s, padding=1), 0.1)
        assert (igemm.to_nchw_float(first, cout) - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("case", [  # (N, Cin, H, W, Cout, k, act)
    (2, 64, 33, 50, 128, 3, igemm.ACT_LEAKY), (1, 128, 40, 70, 64, 3, igemm.ACT_RELU), (1, 96, 19, 33, 70, 3, igemm.ACT_NONE),
    (1, 64, 37, 61, 32, 3, igemm.ACT_RELU), (1, 64, 19, 33, 64, 5, igemm.ACT_RELU), (1, 64, 41, 77, 32, 7, igemm.ACT_RELU),
    (2, 256, 68, 120, 256, 3, igemm.ACT_RELU)])
def test_patch_kernel_with_weights_in_lds(case):
    """k_conv_patch_lw (the chunk's weight block staged in LDS once per workgroup, 64 out-channels per workgroup for 3x3) against
    the stock operator and against k_conv_patch_r8 (weight fragments per wave from L2): the same loop nest per accumulator, so
    the same bits."""
    from video_super_resolution_amd import _lib as L
    N, cin, H, W, cout, k, act = case
    rs = np.random.RandomState(cin + 17 * cout + k)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=1, padding=k // 2)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=1, pad=k // 2, act=act)
    lib = L.load()
    xs = igemm.to_nhwc_half(x)
    try:
        lib.vsr_conv2d_tuning(2)        # every legal layer through the patch builds
        lib.vsr_conv2d_tuning(2000)     # (not the tile kernel)
        lib.vsr_conv2d_tuning(6002)
        out = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("patch_lw<"), lib.vsr_last_route().decode()
        lib.vsr_conv2d_tuning(6000)
        out_r8 = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("patch_r8<"), lib.vsr_last_route().decode()
    finally:
        lib.vsr_conv2d_tuning(0)
        lib.vsr_conv2d_tuning(2001)
        lib.vsr_conv2d_tuning(6001)
    got = igemm.to_nchw_float(out, cout)
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    if out.shape[3] > cout:
        assert float(out[..., cout:].abs().max()) == 0.0
    assert torch.equal(out, out_r8)


@pytest.mark.parametrize("case", [  # (N, H, W, Cout, act, out_ld extra, out_coff)
    (1, 259, 271, 208, igemm.ACT_RELU, 0, 0), (2, 190, 181, 128, igemm.ACT_LEAKY, 0, 0), (1, 300, 230, 112, igemm.ACT_NONE, 0, 0),
    (1, 270, 250, 224, igemm.ACT_RELU, 32, 32), (3, 160, 140, 160, igemm.ACT_RELU, 0, 0)])
def test_conv1x1_transposing_build(case):
    """k_conv1x1_t (128 input channels; contiguous KiB accesses through per-wave LDS slots, LDS-DMA input one block ahead) against
    the stock operator and against k_conv1x1_stream: the same products in the same order per accumulator, so the same bits; ragged
    last blocks, a channel-slice destination."""
    from video_super_resolution_amd import _lib as L
    N, H, W, cout, act, extra, coff = case
    cin = 128
    rs = np.random.RandomState(H + cout)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=1, pad=0, act=act)
    lib = L.load()
    xs = igemm.to_nhwc_half(x)
    ld = igemm.pad32(coff + cout) + extra
    outs = []
    try:
        for mode in (7001, 7000):
            lib.vsr_conv2d_tuning(mode)
            dst = torch.full((N, H, W, ld), 3.0, dtype=torch.float16, device="cuda")
            conv(xs, out=dst, out_coff=coff)
            outs.append((dst, lib.vsr_last_route().decode()))
    finally:
        lib.vsr_conv2d_tuning(7000)   # (the default: the streaming build; LAB_NOTES.md 5.3)
    (new, r_new), (old, r_old) = outs
    assert r_new == "conv1x1_t" and r_old.startswith("conv1x1_stream<4>"), (r_new, r_old)
    got = new[..., coff:coff + cout].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert torch.equal(new, old)                                   # incl. the untouched channels outside the slice (3.0)
    assert float((new[..., :coff] - 3.0).abs().max() if coff else 0.0) == 0.0


@pytest.mark.parametrize("case", [  # (N, Cin, H, W, Cout, k, stride, pad, act)
    (1, 64, 17, 30, 64, 3, 1, 1, igemm.ACT_LEAKY), (1, 512, 9, 15, 512, 3, 1, 1, igemm.ACT_LEAKY), (2, 256, 34, 30, 512, 3, 2, 1, igemm.ACT_RELU),
    (1, 1024, 9, 15, 1024, 3, 1, 1, igemm.ACT_LEAKY), (1, 96, 23, 19, 48, 5, 1, 2, igemm.ACT_NONE), (1, 32, 40, 56, 16, 3, 1, 1, igemm.ACT_RELU),
    (1, 128, 12, 10, 128, 7, 2, 3, igemm.ACT_RELU), (1, 64, 5, 7, 96, 1, 1, 0, igemm.ACT_NONE)])
def test_gather_kernel_five_set_ring(case):
    """k_conv_igemm_d with five register sets (four K steps in flight; launches of at most one workgroup per CU) against the
    stock operator and against the three-set build: the same products in the same order, so the same bits -- K ranges that are
    not multiples of five or three, split K, the tail pair, fewer K steps than sets."""
    from video_super_resolution_amd import _lib as L
    N, cin, H, W, cout, k, stride, pad, act = case
    rs = np.random.RandomState(cin + cout + k)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=stride, padding=pad)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=stride, pad=pad, act=act)
    lib = L.load()
    xs = igemm.to_nhwc_half(x)
    outs = []
    try:
        lib.vsr_conv2d_tuning(2000)      # (the gather kernel on every layer)
        old_patch = lib.vsr_conv2d_tuning(1)     # (and no patch kernels)
        for mode in (8002, 8000):
            lib.vsr_conv2d_tuning(mode)
            outs.append((conv(xs).clone(), lib.vsr_last_route().decode()))
    finally:
        lib.vsr_conv2d_tuning(8000)      # (the default)
        lib.vsr_conv2d_tuning(2001)
        lib.vsr_conv2d_tuning(old_patch)
    (new, r_new), (old, r_old) = outs
    assert "gather" in r_new and "gather" in r_old, (r_new, r_old)
    got = new[..., :cout].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() <= 3e-3 * ref.abs().max().item()
    assert torch.equal(new, old)


@pytest.mark.parametrize("mode", [9002, 9003])
@pytest.mark.parametrize("case", [  # (N, Cin, H, W, Cout, k, act)
    (2, 64, 37, 61, 16, 3, igemm.ACT_RELU),     # the hourglass's thin 3x3 (two chunks), ragged tiles
    (3, 64, 33, 50, 1, 3, igemm.ACT_NONE),      # its final conv (one live out-channel)
    (1, 224, 40, 70, 2, 3, igemm.ACT_NONE),     # predict_flow over 7 chunks
    (2, 32, 19, 33, 32, 3, igemm.ACT_RELU),     # one chunk: the pipeline runs across tiles only
    (1, 96, 50, 97, 70, 3, igemm.ACT_LEAKY),    # odd out-channel count (cout_pad 128: 64 / 32 per workgroup by mode)
    (2, 256, 35, 64, 256, 3, igemm.ACT_RELU),   # VGG-stage shape: 8 chunks x 4 out-channel blocks
    (1, 32, 23, 45, 32, 5, igemm.ACT_RELU), (2, 64, 41, 33, 16, 5, igemm.ACT_RELU),
    (4, 32, 300, 210, 64, 3, igemm.ACT_RELU),   # more work items than resident workgroups: every workgroup walks several tiles
])
def test_patch_kernel_persistent_prefetching(case, mode):
    """k_conv_patch_pf (conv_patch_pf.hip: persistent over (tile, out-channel block) items, the next stage's patch and weight block
    prefetched into registers before the tap walk) against the stock operator and against k_conv_patch_r8: the same loop nest per
    accumulator, so the same bits -- with 64 (mode 9002) / at most 32 (9003) out-channels per workgroup."""
    from video_super_resolution_amd import _lib as L
    N, cin, H, W, cout, k, act = case
    rs = np.random.RandomState(cin + 17 * cout + k + H)
    x = torch.from_numpy(rs.randn(N, cin, H, W).astype(np.float32)).cuda().half()
    w = torch.from_numpy((rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float32)).cuda().half().float()
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).cuda()
    ref = F.conv2d(x.float(), w, b, stride=1, padding=k // 2)
    ref = F.relu(ref) if act == igemm.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == igemm.ACT_LEAKY else ref)
    conv = igemm.HConv(w, b, stride=1, pad=k // 2, act=act)
    lib = L.load()
    xs = igemm.to_nhwc_half(x)
    try:
        lib.vsr_conv2d_tuning(2)        # every legal layer through the patch builds
        lib.vsr_conv2d_tuning(2000)     # (not the tile kernel)
        lib.vsr_conv2d_tuning(mode)
        out = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("patch_pf<"), lib.vsr_last_route().decode()
        lib.vsr_conv2d_tuning(9000)
        lib.vsr_conv2d_tuning(6000)
        out_r8 = conv(xs).clone()
        assert lib.vsr_last_route().decode().startswith("patch_r8<"), lib.vsr_last_route().decode()
    finally:
        lib.vsr_conv2d_tuning(0)
        lib.vsr_conv2d_tuning(2001)
        lib.vsr_conv2d_tuning(6001)
        lib.vsr_conv2d_tuning(9001)
    got = igemm.to_nchw_float(out, cout)
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    if out.shape[3] > cout:
        assert float(out[..., cout:].abs().max()) == 0.0
    assert torch.equal(out, out_r8)


@pytest.mark.parametrize("case", [  # (N, cin, H, W, in_ld extra, with upsampling, upsampling bias)
    (2, 1024, 8, 15, 0, True, True), (2, 1026, 16, 30, 32, True, False), (1, 770, 32, 60, 32, True, True), (2, 386, 64, 120, 32, True, False),
    (2, 194, 128, 240, 32, False, False), (1, 128, 37, 45, 0, True, True), (1, 32, 21, 50, 0, True, True), (3, 16, 9, 7, 16, False, False),
    (1, 512, 5, 3, 0, True, False)])
def test_flow_head_matches_torch(case):
    """igemm.HFlowHead (csrc/conv_flow_head.hip: predict_flow as a 1x1 convolution onto 18 tap-channels + the shifted sum, the flow
    upsampling ConvTranspose2d(2, 2, 4, 2, 1) fused behind it) against the stock operators on the same fp16-rounded operands: every
    FlowNet head geometry (8 x 15 ... 128 x 240; the concat buffers' odd channel counts; with / without the upsampling and its bias),
    ragged tiles, one- and many-chunk contractions.  The flow is compared at the fp16 bar; the upsampled flow against the stock
    transposed convolution OF THE KERNEL'S OWN fp16 flow (what the unfused path computes)."""
    N, cin, H, W, extra, with_up, up_bias = case
    rs = np.random.RandomState(cin + H)
    ld = igemm.pad32(cin) + extra
    x = torch.zeros((N, H, W, ld), dtype=torch.float16, device="cuda")
    x[..., :cin] = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)).cuda().half()
    if extra:
        x[..., igemm.pad32(cin):] = 7.0    # channels beyond the slice must not be read
    wp = torch.from_numpy((rs.randn(2, cin, 3, 3) / np.sqrt(9 * cin)).astype(np.float32)).cuda().half().float()
    bp = torch.from_numpy(rs.randn(2).astype(np.float32)).cuda()
    wu = torch.from_numpy((rs.randn(2, 2, 4, 4) * 0.3).astype(np.float32)).cuda().half().float() if with_up else None
    bu = torch.from_numpy(rs.randn(2).astype(np.float32)).cuda() if (with_up and up_bias) else None
    head = igemm.HFlowHead(wp, bp, wu, bu)
    up_out, coff = None, 0
    if with_up:
        coff = 6
        up_out = torch.full((N, 2 * H, 2 * W, 32), 3.0, dtype=torch.float16, device="cuda")
    flow = head(x, up_out=up_out, up_coff=coff)
    ref = F.conv2d(x[..., :cin].permute(0, 3, 1, 2).float(), wp, bp, padding=1)
    got = flow[..., :2].permute(0, 3, 1, 2).float()
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert float(flow[..., 2:].abs().max()) == 0.0
    if with_up:
        ref_up = F.conv_transpose2d(got, wu, bu, stride=2, padding=1)
        got_up = up_out[..., coff:coff + 2].permute(0, 3, 1, 2).float()
        assert (got_up - ref_up).abs().max().item() <= 2e-3 * max(ref_up.abs().max().item(), 1e-3)
        assert float((up_out[..., :coff] - 3.0).abs().max()) == 0.0 and float((up_out[..., coff + 2:] - 3.0).abs().max()) == 0.0
    first = flow.clone()
    again = head(x, up_out=up_out, up_coff=coff)
    assert torch.equal(again, first)     # deterministic (the four waves' partial sums are added in wave order)
