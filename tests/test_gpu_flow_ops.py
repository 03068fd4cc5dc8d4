"""HIP kernels of FlowNet2's native operators and the flow colour coding against the oracle, through the C ABI.

Tolerances: resample2d / channelnorm / the fused warp kernels restate the reference's float-double arithmetic
without FMA contraction, so they are compared BIT-EXACTLY with oracle/native_ops.c; correlation sums 256 fp32
products in a different (documented) order than the reference's 32-lane tree, bar 1e-5 of the value range;
flow2img produces uint8 values and is compared exactly.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import native, vsr_oracle as O  # noqa: E402
from video_super_resolution_amd import ops  # noqa: E402


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape,scale", [((2, 3, 40, 56), 8.0), ((1, 3, 64, 128), 1.5), ((1, 5, 7, 301), 300.0),
                                        ((1, 1, 1, 1), 0.3)])
def test_resample2d_bit_exact(shape, scale):
    rs = np.random.RandomState(sum(shape))
    B, C, H, W = shape
    img = rs.randn(*shape).astype(np.float32)
    flow = (rs.randn(B, 2, H, W) * scale).astype(np.float32)
    flow[0, 0, 0, 0] = 1e9      # far outside: both indices clamp to the border
    flow[0, 1, -1, -1] = -1e9
    out = ops.resample2d(_cuda(img), _cuda(flow)).cpu().numpy()
    np.testing.assert_array_equal(out, native.resample2d(img, flow))
    out_n = ops.resample2d(_cuda(img), _cuda(flow), bilinear=False).cpu().numpy()
    np.testing.assert_array_equal(out_n, native.resample2d(img, flow, bilinear=False))


def test_resample2d_rejects_cpu_tensors_and_bad_kernel():
    from video_super_resolution_amd._lib import VsrHipError
    with pytest.raises(VsrHipError):
        ops.resample2d(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(VsrHipError):
        ops.resample2d(torch.zeros(1, 3, 4, 4).cuda(), torch.zeros(1, 2, 4, 4).cuda(), kernel_size=3)


@pytest.mark.parametrize("shape", [(2, 3, 17, 19), (1, 2, 64, 128), (3, 7, 5, 1)])
def test_channelnorm_bit_exact(shape):
    x = np.random.RandomState(7).randn(*shape).astype(np.float32)
    np.testing.assert_array_equal(ops.channelnorm(_cuda(x)).cpu().numpy(), native.channelnorm(x))


@pytest.mark.parametrize("shape", [(1, 256, 8, 16), (2, 64, 12, 14), (1, 40, 5, 37)])
def test_correlation(shape):
    rs = np.random.RandomState(11)
    f1, f2 = rs.randn(*shape).astype(np.float32), rs.randn(*shape).astype(np.float32)
    ref = native.correlation(f1, f2, 20, 1, 20, 1, 2)
    out = ops.correlation(_cuda(f1), _cuda(f2), 20, 1, 20, 1, 2).cpu().numpy()
    assert out.shape == ref.shape == (shape[0], 441, shape[2], shape[3])
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5 * np.abs(ref).max())


def test_correlation_module_matches_reference_api():
    m = ops.Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
    a = torch.randn(1, 32, 6, 9, device="cuda")
    out = m(a, a)
    # zero displacement channel (index 220) of a tensor with itself is the mean of squares
    torch.testing.assert_close(out[:, 220], (a * a).mean(1), rtol=1e-5, atol=1e-6)


def test_fused_warp_concat_and_norms_bit_exact():
    rs = np.random.RandomState(5)
    x6 = rs.randn(2, 6, 33, 47).astype(np.float32)
    flow = (rs.randn(2, 2, 33, 47) * 6).astype(np.float32)
    warped = native.resample2d(x6[:, 3:], flow)
    ndiff = native.channelnorm(x6[:, :3] - warped)
    ref12 = np.concatenate([x6, warped, flow / np.float32(20.0), ndiff], 1)
    out12 = ops.warp_concat(_cuda(x6), _cuda(flow), 20.0).cpu().numpy()
    np.testing.assert_array_equal(out12[:, :9], ref12[:, :9])
    np.testing.assert_allclose(out12[:, 9:11], ref12[:, 9:11], rtol=1e-6)  # x*(1/20) vs x/20
    np.testing.assert_array_equal(out12[:, 11:], ref12[:, 11:])
    nf, nd = ops.warp_norms(_cuda(x6), _cuda(flow))
    np.testing.assert_array_equal(nf.cpu().numpy(), native.channelnorm(flow))
    np.testing.assert_array_equal(nd.cpu().numpy(), ndiff)


def test_flow2img_matches_reference_golden(golden):
    g = golden("g3_flow2img")
    for case in ("rand", "zero", "tiny", "nan_unknown", "unknown"):
        fl = g[case + "_in"]
        out = ops.flow2img(_cuda(fl.transpose(2, 0, 1))).cpu().numpy()
        assert out.dtype == np.float32 and out.shape == fl.shape[:2] + (3,)
        np.testing.assert_array_equal(out.astype(np.uint8), g[case + "_out"], err_msg=case)


def test_flow2img_large_random_vs_oracle():
    rs = np.random.RandomState(9)
    fl = (rs.randn(256, 384, 2) * rs.choice([0.01, 1.0, 30.0], size=(256, 384, 1))).astype(np.float32)
    out = ops.flow2img(_cuda(fl.transpose(2, 0, 1))).cpu().numpy().astype(np.uint8)
    ref = O.flow2img(fl.copy())
    # float64 atan2 of the device library vs libm may differ in the last ulp: allow isolated +-1 steps
    diff = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-4


# ------------------------------------------------------------------------------------------------------------------
# Round 2: the frame glue fused around the kernels above, against the stock-op compositions they replace.
def test_prepare_pairs_matches_flownet2_input_normalisation():
    """models.py:74-79 + StaticCenterCrop: x = (inputs - rgb_mean) / 255 for pairs picked from a frame stack."""
    import ctypes
    from video_super_resolution_amd import _lib as L
    rs = np.random.RandomState(11)
    frames = torch.from_numpy(rs.randint(0, 256, (3, 70, 134, 3)).astype(np.float32)).cuda()
    pairs, (y0, x0, H, W) = [(0, 1), (1, 2)], (3, 3, 64, 128)
    B = len(pairs)
    x = torch.empty((B, 6, H, W), device="cuda")
    x6 = torch.empty((B, H, W, 32), dtype=torch.float16, device="cuda")
    both = torch.empty((2 * B, H, W, 4), dtype=torch.float16, device="cuda")
    ws = torch.empty(B * 128 * 3, device="cuda")
    L.check(L.load().vsr_flownet_prepare_pairs(L.dptr(frames), 3, 70, 134, (ctypes.c_int * B)(0, 1), (ctypes.c_int * B)(1, 2), B, y0, x0,
                                               H, W, L.dptr(ws), L.dptr(x), L.dptr(x6, torch.float16), L.dptr(both, torch.float16),
                                               L.stream()))
    crop = frames[:, y0:y0 + H, x0:x0 + W]
    inputs = torch.stack([torch.stack([crop[a], crop[b]]).permute(3, 0, 1, 2) for a, b in pairs])       # [B,3,2,H,W]
    mean = inputs.contiguous().view(B, 3, -1).mean(-1).view(B, 3, 1, 1, 1)
    ref = (inputs - mean) / 255.0
    ref = torch.cat((ref[:, :, 0], ref[:, :, 1]), 1)
    assert (x - ref).abs().max().item() < 2e-6
    assert torch.equal(x6[..., :6], x.permute(0, 2, 3, 1).half()) and not x6[..., 6:].any()
    assert torch.equal(both[:B, ..., :3], x[:, :3].permute(0, 2, 3, 1).half()) and torch.equal(both[B:, ..., :3], x[:, 3:].permute(0, 2, 3, 1).half())
    assert not both[..., 3].any()


@pytest.mark.parametrize("bilinear", [1, 0])
def test_fused_flow_epilogues_match_their_compositions(bilinear):
    import torch.nn.functional as F
    from video_super_resolution_amd import _lib as L, ops
    rs = np.random.RandomState(12 + bilinear)
    B, H, W = 2, 64, 96
    x = torch.from_numpy(rs.randn(B, 6, H, W).astype(np.float32) * 0.3).cuda()
    f2 = torch.zeros((B, H // 4, W // 4, 32), dtype=torch.float16, device="cuda")
    f2[..., :2] = torch.from_numpy((rs.randn(B, H // 4, W // 4, 2) * 0.4).astype(np.float16)).cuda()
    g2 = torch.zeros_like(f2)
    g2[..., :2] = torch.from_numpy((rs.randn(B, H // 4, W // 4, 2) * 30).astype(np.float16)).cuda()
    nchw = lambda t: t[..., :2].permute(0, 3, 1, 2).float()
    # ---- upsample x div_flow -> warp -> concat (models.py:83-91)
    out16 = torch.empty((B, H, W, 16), dtype=torch.float16, device="cuda")
    L.check(L.load().vsr_flownet_up_warp_concat16_f16(L.dptr(x), L.dptr(f2, torch.float16), 32, bilinear, L.cf(20.0), L.cf(1 / 20.0),
                                                      L.dptr(out16, torch.float16), B, H, W, L.stream()))
    flow = F.interpolate(nchw(f2), scale_factor=4, mode="bilinear" if bilinear else "nearest") * 20.0
    ref = ops.warp_concat(x, flow, 20.0).permute(0, 2, 3, 1)                         # [B,H,W,12]
    assert (out16[..., :12].float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert not out16[..., 12:].any()
    # ---- the fusion network's input (models.py:106-125)
    out32 = torch.empty((B, H, W, 32), dtype=torch.float16, device="cuda")
    L.check(L.load().vsr_flownet_fusion_input_f16(L.dptr(x), L.dptr(g2, torch.float16), 32, L.dptr(f2, torch.float16), 32, L.cf(20.0),
                                                  L.dptr(out32, torch.float16), B, H, W, L.stream()))
    fsd = F.interpolate(nchw(g2), scale_factor=4, mode="nearest") / 20.0
    fs2 = F.interpolate(nchw(f2), scale_factor=4, mode="nearest") * 20.0
    n_sd, d_sd = ops.warp_norms(x, fsd)
    n_s2, d_s2 = ops.warp_norms(x, fs2)
    ref = torch.cat((x[:, :3], fsd, fs2, n_sd, n_s2, d_sd, d_s2), 1).permute(0, 2, 3, 1)
    assert torch.equal(out32[..., :11], ref.half())          # nearest upsampling and the warps are the same arithmetic: exact
    assert not out32[..., 11:].any()


def test_flow2img_from_nhwc_half_equals_planar():
    from video_super_resolution_amd import ops
    rs = np.random.RandomState(13)
    m = torch.zeros((40, 56, 32), dtype=torch.float16, device="cuda")
    m[..., :2] = torch.from_numpy((rs.randn(40, 56, 2) * 5).astype(np.float16)).cuda()
    assert torch.equal(ops.flow2img_nhwc(m), ops.flow2img(m[..., :2].permute(2, 0, 1).float()))


@pytest.mark.parametrize("first_call", [True, False])
def test_plane_assembly_matches_the_reference_expressions(first_call):
    """video_super_resolution.py:33-40 / :57-62 as stock ops vs the one-launch assembly (byte moves and (a+b)/2: exact)."""
    import torch.nn.functional as F
    from video_super_resolution_amd import _lib as L
    from video_super_resolution_amd.vsr import VSR, maskprocess
    rs = np.random.RandomState(14)
    h, w, Hc, Wc = 66, 70, 64, 64
    d = torch.from_numpy(rs.randint(0, 256, (3, h, w, 3)).astype(np.float32)).cuda()
    pics = torch.from_numpy(rs.randint(0, 256, (2, Hc, Wc, 3)).astype(np.float32)).cuda()
    z = [torch.from_numpy(rs.randn(1, 1, h, w).astype(np.float32)).cuda() for _ in range(3)]
    frames = d.permute(0, 3, 1, 2)
    depth = torch.stack([maskprocess(torch.squeeze(torch.mean(torch.stack([z[0], z[1]]), 0)[0])),
                         maskprocess(torch.squeeze(torch.mean(torch.stack([z[1], z[2]]), 0)[0]))])
    fl = F.interpolate(pics.permute(0, 3, 1, 2), (h, w))
    if first_call:
        ref = torch.cat((frames, fl, depth, frames[0:1]), 0)
        got = VSR._assemble(d, pics, z)
    else:
        prev = torch.from_numpy(rs.randint(0, 256, (1, 4 * h, 4 * w, 3)).astype(np.float32)).cuda()
        est = torch.empty((3, h, w), device="cuda")
        est_hw3 = torch.empty((h, w, 3), device="cuda")
        L.check(L.load().vsr_resize_estimate_f32(L.dptr(prev), 4 * h, 4 * w, L.dptr(est), L.dptr(est_hw3), h, w, L.stream()))
        want = F.interpolate(prev.permute(0, 3, 1, 2), (h, w))
        assert torch.equal(est, want[0]) and torch.equal(est_hw3, want[0].permute(1, 2, 0))
        mask = (torch.from_numpy(rs.rand(h, w).astype(np.float32)).cuda() > 0.5).float()
        masked = torch.where(maskprocess(mask) != 0, torch.zeros_like(est), est).unsqueeze(0)
        ref = torch.cat((frames, fl, depth, masked), 0)
        got = VSR._assemble(d, pics, z, est, mask)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 36, 200), (2, 132, 76), (1, 512, 960)])
@pytest.mark.parametrize("bilinear", [1, 0])
@pytest.mark.parametrize("sigma", [0.05, 0.6, 8.0])
def test_lds_staged_warp_equals_the_gather_build(shape, bilinear, sigma):
    """The warp of vsr_flownet_up_warp_concat16_f16 as BASELINE.json's north_star words it (LDS-staged source tile with a halo of
    8 pixels, the right column of each lane's 2x2 neighbourhood handed over from the next lane by DPP, global fallback beyond the
    halo) against the thread-per-pixel gather build: the same values into the same arithmetic (resample2d_kernel.cu:16-72), bit for
    bit -- flows of a fraction of a pixel (every lane takes its neighbour's column), of ~12 pixels (mixed) and of ~160 pixels
    (fallback, clamping at the borders), sizes that are not multiples of the 4 x 64 tile."""
    from video_super_resolution_amd import _lib as L
    B, H, W = shape
    rs = np.random.RandomState(H + W + bilinear)
    x = torch.from_numpy(rs.rand(B, 6, H, W).astype(np.float32)).cuda()
    f2 = torch.zeros((B, H // 4, W // 4, 32), dtype=torch.float16, device="cuda")
    f2[..., :2] = torch.from_numpy((rs.randn(B, H // 4, W // 4, 2) * sigma).astype(np.float16)).cuda()
    if sigma > 1:
        f2[0, 0, 0, 0] = float("nan")      # a NaN flow: float -> int saturates / NaN -> 0, the clamps keep the index in range
        f2[0, 1, 1, 1] = 60000.0
    lib = L.load()
    outs = []
    try:
        for variant in (1, 0):
            L.check(lib.vsr_flownet_warp_variant(variant))
            out16 = torch.full((B, H, W, 16), 7.0, dtype=torch.float16, device="cuda")
            L.check(lib.vsr_flownet_up_warp_concat16_f16(L.dptr(x), L.dptr(f2, torch.float16), 32, bilinear, L.cf(20.0), L.cf(1 / 20.0),
                                                         L.dptr(out16, torch.float16), B, H, W, L.stream()))
            outs.append(out16)
    finally:
        lib.vsr_flownet_warp_variant(1)   # (the default: the gather build is the faster one, LAB_NOTES.md 5.4)
    a, b = outs
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    assert bool(same.all()), int((~same).sum())
