"""SRProjectionModule on the GPU (hand-written HIP kernels through the C ABI) against the golden vectors of
the reference and against the oracle.

Tolerance (BASELINE.json north_star): 1e-3 relative, fp32.  "Relative" is taken against the value range of the
compared tensor (max |reference|): the exact-fp32 device path differs from ATen's CPU convolutions only by
summation order, so it is held to 2e-5 of the range -- 50x tighter than the stated bar.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vsr_oracle as O  # noqa: E402

TOL_FP32 = 2e-5


def _close(a, ref, tol, what=""):
    ref = np.asarray(ref)
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    err = np.abs(a - ref).max()
    assert err <= tol * np.abs(ref).max(), f"{what}: max err {err:.3e} vs range {np.abs(ref).max():.3e}"


@pytest.mark.parametrize("tag", ["16x16", "12x20"])
def test_sr_matches_reference_golden(golden, gpu_vsr, tag):
    g = golden(f"g1_sr_{tag}")
    taps = {}
    out = gpu_vsr.model(torch.from_numpy(g["x"]).cuda(), taps=taps)
    _close(taps["feat_in"], g["feat_in"], TOL_FP32, "feat_in")
    for s in range(3):
        _close(taps[f"block{s}"], g[f"block{s}"], TOL_FP32, f"block{s}")
    _close(taps["prefc2"], g["prefc2"], TOL_FP32, "prefc2")
    _close(out, g["out"], TOL_FP32, "out")


def test_sr_group_tensors(golden, gpu_vsr):
    g = golden("g2_groups")
    taps = {}
    out = gpu_vsr.model(torch.from_numpy(g["x"]).cuda(), taps=taps)
    _close(out, g["out"], TOL_FP32, "out")
    _close(taps["lr0"], g["lr0"], TOL_FP32, "lr0")
    _close(taps["lr3"], g["lr3"], TOL_FP32, "lr3")
    _close(taps["lr6"], g["lr6"], TOL_FP32, "lr6")


@pytest.mark.parametrize("hw", [(5, 7), (9, 33), (2, 2), (31, 17)])
def test_sr_ragged_sizes_vs_oracle(gpu_vsr, oracle_params, hw):
    """Sizes that are not multiples of anything, down to 2x2 (the reference's `.squeeze()` needs h,w > 1)."""
    rs = np.random.RandomState(hw[0] * 100 + hw[1])
    x = torch.from_numpy(rs.randint(0, 256, (8, 3) + hw).astype(np.float32))
    P = {k[len("model."):]: v for k, v in oracle_params.items() if k.startswith("model.")}
    with torch.no_grad():
        ref = O.sr_forward(P, x)
    _close(gpu_vsr.model(x.cuda()), ref.numpy(), TOL_FP32, f"{hw}")


def test_sr_weight_update_invalidates_caches(gpu_vsr, oracle_params):
    import copy
    m = copy.deepcopy(gpu_vsr.model)
    x = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (8, 3, 6, 6)).astype(np.float32)).cuda()
    a = m(x)
    with torch.no_grad():
        m.block.downBlocks[0][0].bias.add_(0.5)  # only reaches the output through the cached constant branch
    b = m(x)
    assert not torch.equal(a, b)
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.sr_forward(P, x.cpu())
    _close(b, ref.numpy(), TOL_FP32, "after update")


def test_sr_properties_at_headline_width(gpu_vsr):
    """Size-independent properties on a full-width strip (LR 24x960 of the 540x960 headline frame): the run is
    deterministic, and the trunk treats the 8 planes independently, so permuting input planes permutes the
    pre-fusion planes."""
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randint(0, 256, (8, 3, 24, 960)).astype(np.float32)).cuda()
    t1, t2, t3 = {}, {}, {}
    o1 = gpu_vsr.model(x, taps=t1)
    o2 = gpu_vsr.model(x, taps=t2)
    assert torch.equal(o1, o2) and torch.equal(t1["prefc2"], t2["prefc2"])
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device="cuda")
    gpu_vsr.model(x[perm].contiguous(), taps=t3)
    assert torch.equal(t3["prefc2"], t1["prefc2"][perm])
    assert o1.shape == (1, 3, 96, 3840) and torch.isfinite(o1).all() and (o1 >= 0).all()


@pytest.mark.parametrize("scale", [4, 2, 3])
@pytest.mark.parametrize("shape", [(2, 9, 40), (1, 37, 33), (3, 5, 70), (1, 1, 1)])
def test_f32_mfma_builds_bit_identical(scale, shape):
    """The float32 ConvTranspose2d / Conv2d (32, 32, K, S, p2) + PReLU blocks of the FeedbackBlock on v_mfma_f32_32x32x2_f32
    (csrc/sr_f32_mfma.hip, the default) against the one-pixel-per-thread kernels (csrc/sr_f32.hip): the same fused multiply-adds
    in the same order -> equal maps; and against the stock operators at the float32 bar.  Ragged widths (column tiles of 32 / 64),
    heights that are not multiples of the 4 rows of a workgroup, slopes above one."""
    import torch.nn.functional as F
    from video_super_resolution_amd import _lib as L
    K = {4: 8, 2: 6, 3: 7}[scale]
    N, h, w = shape
    rs = np.random.RandomState(scale * 100 + h + w)
    lib = L.load()
    lib.vsr_sr_f32_variant.restype = __import__("ctypes").c_int
    x = torch.from_numpy(rs.randn(N, 32, h, w).astype(np.float32)).cuda()
    wt_d = torch.from_numpy((rs.randn(32, 32, K, K) / (4.0 * K)).astype(np.float32)).cuda()   # ConvTranspose2d weight [in, out, K, K]
    wt_c = torch.from_numpy((rs.randn(32, 32, K, K) / (4.0 * K)).astype(np.float32)).cuda()   # Conv2d weight [out, in, K, K]
    b = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    slope = 1.3 if h == 5 else 0.2
    wp_d = wt_d.permute(2, 3, 0, 1).contiguous()      # [ky][kx][in][out]
    wp_c = wt_c.permute(2, 3, 1, 0).contiguous()
    outs = []
    try:
        for variant in (0, 1, 2):
            lib.vsr_sr_f32_variant(variant)
            hr = torch.empty((N, 32, scale * h, scale * w), dtype=torch.float32, device="cuda")
            L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp_d), L.dptr(b), L.cf(slope), L.dptr(hr), N, h, w, scale, None, None, L.cf(0.0), L.stream()), "deconv")
            lr = torch.empty((N, 32, h, w), dtype=torch.float32, device="cuda")
            L.check(lib.vsr_sr_conv_f32(L.dptr(hr), L.dptr(wp_c), L.dptr(b), L.cf(slope), L.dptr(lr), N, h, w, scale, L.stream()), "conv")
            outs.append((hr, lr))
    finally:
        lib.vsr_sr_f32_variant(0)
    (hr0, lr0), (hr1, lr1), (hr2, lr2) = outs
    assert torch.equal(hr0, hr1) and torch.equal(lr0, lr1) and torch.equal(hr0, hr2) and torch.equal(lr0, lr2)
    a = torch.tensor([slope], device="cuda")
    hr_ref = F.prelu(F.conv_transpose2d(x.double(), wt_d.double(), b.double(), stride=scale, padding=2), a.double())
    lr_ref = F.prelu(F.conv2d(hr0.double(), wt_c.double(), b.double(), stride=scale, padding=2), a.double())
    _close(hr0.double(), hr_ref.cpu().numpy(), TOL_FP32, "deconv")
    _close(lr0.double(), lr_ref.cpu().numpy(), TOL_FP32, "conv")


@pytest.mark.parametrize("case", [(2, 9, 13, 2), (1, 20, 70, 2), (1, 7, 40, 4), (1, 6, 33, 3)])
def test_f32_deconv_with_fused_downtran_matches_the_two_launches(case):
    """vsr_sr_deconv_f32 with the downtran 1x1 + PReLU in its epilogue (dt_frags) against the deconvolution followed by
    vsr_sr_conv1x1_f32: the same products, the 1x1's sum in another order -> 1e-6 of the range, and the float32 bar against float64."""
    import ctypes
    from video_super_resolution_amd import _lib as L
    from video_super_resolution_amd.sr import pack_dt_frags
    N, h, w, S = case
    K = {2: 6, 3: 7, 4: 8}[S]
    rs = np.random.RandomState(h * 3 + w + S)
    lib = L.load()
    x = torch.from_numpy(rs.randn(N, 32, h, w).astype(np.float32)).cuda()
    wt = torch.from_numpy((rs.randn(32, 32, K, K) / (4.0 * K)).astype(np.float32)).cuda()       # ConvTranspose2d weight [in, out, K, K]
    wp = wt.permute(2, 3, 0, 1).contiguous()
    b = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    wdt = torch.from_numpy((rs.randn(32, 32 * 3) / 6.0).astype(np.float32)).cuda()              # downtran matrix [out, ld]; this map reads columns 32 ..
    bdt = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    fr = pack_dt_frags(wdt, 32)
    fused = torch.empty((N, 32, S * h, S * w), dtype=torch.float32, device="cuda")
    L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(fused), N, h, w, S, L.dptr(fr), L.dptr(bdt), L.cf(0.3), L.stream()), "deconv+dt")
    hr = torch.empty_like(fused)
    L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(hr), N, h, w, S, None, None, L.cf(0.0), L.stream()), "deconv")
    two = torch.empty_like(fused)
    P = S * S * h * w
    L.check(lib.vsr_sr_conv1x1_f32(L.dptr(hr), ctypes.c_void_p(wdt.data_ptr() + 4 * 32), wdt.shape[1], L.optr(None), L.optr(None), 0, L.optr(None), L.optr(None), 0,
                                   L.dptr(bdt), L.optr(None), L.cf(0.3), L.dptr(two), N, P, L.stream()), "conv1x1")
    rng = two.abs().max().item()
    assert (fused - two).abs().max().item() <= 1e-6 * rng
    ref = F.prelu(F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=S, padding=2), torch.tensor([0.2], device="cuda").double())
    ref = torch.einsum("oc,nchw->nohw", wdt[:, 32:64].double(), ref) + bdt.double()[None, :, None, None]
    ref = torch.where(ref >= 0, ref, 0.3 * ref)
    _close(fused.double(), ref.cpu().numpy(), TOL_FP32, "deconv+dt")


def test_f32_forward_fused_downtran_matches_separate_launches():
    """SRProjectionModule (float32, x2 and x4): the forward with the downtran 1x1 fused into the deconvolution (default) against
    fuse_dt_f32 = False."""
    from video_super_resolution_amd import SRProjectionModule
    from video_super_resolution_amd.weights import fill_module_
    for scale in (2, 4):
        m = fill_module_(SRProjectionModule(upscale_factor=scale).eval(), seed=0, prefix="model.").cuda()
        m.precision = "fp32"
        x = torch.from_numpy(np.random.RandomState(scale).randint(0, 256, (8, 3, 24, 40)).astype(np.float32)).cuda()
        with torch.no_grad():
            a = m(x)
            m.fuse_dt_f32 = False
            b = m(x)
            m.fuse_dt_f32 = True
            m.tail_conv_mfma_f32 = False     # conv_out on the one-pixel-per-thread k_tail instead of the 16x16x4 MFMA kernel
            c = m(x)
            m.tail_conv_mfma_f32 = True
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
        assert (a - c).abs().max().item() <= 2e-6 * c.abs().max().item()


@pytest.mark.parametrize("case", [(2, 1000, 1, False), (1, 4097, 2, True), (3, 70, 3, True), (1, 31, 1, True)])
def test_f32_conv1x1_mfma_bit_identical(case):
    """vsr_sr_conv1x1_f32 on the matrix cores against its one-pixel-per-thread build: one to three inputs with weight slices of
    a wider matrix, the constant map, ragged pixel counts."""
    import ctypes
    from video_super_resolution_amd import _lib as L
    N, P, nin, with_map = case
    rs = np.random.RandomState(P + nin)
    lib = L.load()
    ins = [torch.from_numpy(rs.randn(N, 32, P).astype(np.float32)).cuda() for _ in range(nin)]
    wfull = torch.from_numpy((rs.randn(32, 32 * nin + 7) / 8.0).astype(np.float32)).cuda()      # [32, ld]: input i reads columns 32 i + 3 ..
    bias = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    cmap = torch.from_numpy(rs.randn(32, P).astype(np.float32)).cuda() if with_map else None
    ld = wfull.shape[1]
    outs = []
    try:
        for variant in (0, 1):
            lib.vsr_sr_f32_variant(variant)
            out = torch.empty((N, 32, P), dtype=torch.float32, device="cuda")
            args = []
            for i in range(3):
                if i < nin:
                    args += [L.dptr(ins[i]), ctypes.c_void_p(wfull.data_ptr() + 4 * (32 * i + 3)), ld]
                else:
                    args += [L.optr(None), L.optr(None), 0]
            L.check(lib.vsr_sr_conv1x1_f32(*args, L.dptr(bias), L.optr(cmap), L.cf(0.25), L.dptr(out), N, P, L.stream()), "conv1x1")
            outs.append(out)
    finally:
        lib.vsr_sr_f32_variant(0)
    assert torch.equal(outs[0], outs[1])
    ref = bias.double()[None, :, None] + (cmap.double()[None] if with_map else 0.0)
    for i in range(nin):
        ref = ref + torch.einsum("oc,ncp->nop", wfull[:, 32 * i + 3:32 * i + 35].double(), ins[i].double())
    ref = torch.where(ref >= 0, ref, 0.25 * ref)
    _close(outs[0].double(), ref.cpu().numpy(), TOL_FP32, "conv1x1")


def test_f32_shared_planes_bit_identical():
    """float32 path: a second call that shares its first three planes with the first (`shared`) reuses their pre-fusion maps and
    evaluates the network on its other five planes only -- the same frame, bit for bit, as the call on all eight."""
    from video_super_resolution_amd import SRProjectionModule
    from video_super_resolution_amd.weights import fill_module_
    m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), seed=0, prefix="model.").cuda()
    m.precision = "fp32"
    rs = np.random.RandomState(11)
    x1 = torch.from_numpy(rs.randint(0, 256, (8, 3, 20, 36)).astype(np.float32)).cuda()
    x2 = x1.clone()
    x2[3:] = torch.from_numpy(rs.randint(0, 256, (5, 3, 20, 36)).astype(np.float32)).cuda()
    with torch.no_grad():
        shared = {"n": 3}
        a1 = m(x1, shared=shared)
        assert "prefc_f32" in shared
        a2 = m(x2, shared=shared)
        b1, b2 = m(x1), m(x2)
        # the shared planes evaluated AHEAD of both calls, on another stream (what VSR.forward does beside the guidance trunks)
        ahead = {"n": 3}
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            m.precompute_shared(x1[:3].contiguous(), ahead, None)
        torch.cuda.current_stream().wait_stream(st)
        c1, c2 = m(x1, shared=ahead), m(x2, shared=ahead)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert torch.equal(c1, b1) and torch.equal(c2, b2)


@pytest.mark.parametrize("case", [(2, 9, 13), (1, 33, 70), (3, 5, 31)])
def test_f32_head_on_the_matrix_cores_matches_one_pixel_per_thread(case):
    """vsr_sr_head_f32 (sub_mean -> conv_in 3x3 + PReLU -> feat_in 1x1 + PReLU) on v_mfma_f32_32x32x2_f32 against its one-pixel-per-thread
    build (variant 1): stage 1 sums in the same order, stage 2 in register-pair order -> 1e-6 of the range; and the float32 bar against
    float64."""
    from video_super_resolution_amd import _lib as L
    N, h, w = case
    nmid = 128
    rs = np.random.RandomState(h + w)
    lib = L.load()
    lib.vsr_sr_f32_variant.restype = __import__("ctypes").c_int
    x = torch.from_numpy(rs.randint(0, 256, (N, 3, h, w)).astype(np.float32)).cuda()
    ss = torch.from_numpy((rs.rand(3) + 0.5).astype(np.float32)).cuda()
    sb = torch.from_numpy((-100 * rs.rand(3)).astype(np.float32)).cuda()
    w_in = torch.from_numpy((rs.randn(nmid, 27) / 40.0).astype(np.float32)).cuda()
    b_in = torch.from_numpy(rs.randn(nmid).astype(np.float32)).cuda()
    w_feat = torch.from_numpy((rs.randn(32, nmid) / 11.0).astype(np.float32)).cuda()
    b_feat = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    outs = []
    try:
        for variant in (0, 1):
            lib.vsr_sr_f32_variant(variant)
            out = torch.full((N, 32, h, w), float("nan"), dtype=torch.float32, device="cuda")
            L.check(lib.vsr_sr_head_f32(L.dptr(x), L.dptr(ss), L.dptr(sb), L.dptr(w_in), L.dptr(b_in), L.cf(0.2), nmid, L.dptr(w_feat), L.dptr(b_feat),
                                        L.cf(0.3), L.dptr(out), N, h, w, L.stream()), "head")
            outs.append(out)
    finally:
        lib.vsr_sr_f32_variant(0)
    rng = outs[1].abs().max().item()
    assert torch.isfinite(outs[0]).all()
    assert (outs[0] - outs[1]).abs().max().item() <= 1e-6 * rng
    xs = x.double() * ss.double().view(1, 3, 1, 1) + sb.double().view(1, 3, 1, 1)
    f = F.conv2d(xs, w_in.double().view(nmid, 3, 3, 3), b_in.double(), padding=1)
    f = torch.where(f >= 0, f, 0.2 * f)
    ref = F.conv2d(f, w_feat.double().view(32, nmid, 1, 1), b_feat.double())
    ref = torch.where(ref >= 0, ref, 0.3 * ref)
    _close(outs[0].double(), ref.cpu().numpy(), TOL_FP32, "head")
