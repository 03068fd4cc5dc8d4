"""The fp16-storage / fp32-accumulate MFMA path of the SR stack (csrc/sr_f16.hip) on the GPU.

Tolerances.  Every stage stores fp16 (relative rounding 2^-11 = 4.9e-4) and accumulates in fp32, so a fused
up->tran->down stage (two fp16 roundings inside, one at the output) is held to 3e-3 of the output range against
an fp32 evaluation of the same stage, and the whole network (6 such stages per step, 3 steps, 1x1s in between) to
1e-2 of the range on the 32-channel hidden state and 2e-3 of the range / PSNR > 55 dB on the final image, where
the full-precision skip path dominates.  The north_star bar for this configuration is PSNR within 0.05 dB.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _stage_reference(m, j, a_nchw):
    """fp32 stock-op evaluation of  lr[j]' -> up_{j+1} -> downtran_{j+1}[:, slice j+2] -> down_{j+2}."""
    b = m.block
    up, dt, dn = b.upBlocks[j + 1], b.downtranBlocks[j + 1], b.downBlocks[j + 2]
    hr = F.prelu(F.conv_transpose2d(a_nchw, up[0].weight, up[0].bias, stride=4, padding=2), up[1].weight)
    c0 = 32 * (j + 2)
    t = F.prelu(F.conv2d(hr, dt[0].weight[:, c0:c0 + 32], dt[0].bias), dt[1].weight)
    return F.prelu(F.conv2d(t, dn[0].weight, dn[0].bias, stride=4, padding=2), dn[1].weight), hr


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (3, 20, 70), (1, 2, 2), (1, 33, 31), (8, 12, 32)])
@pytest.mark.parametrize("chain", [0, 3])
@pytest.mark.parametrize("kernel", ["roles", "uniform"])
def test_fused_up_tran_down_stage(gpu_vsr_f16, shape, chain, kernel):
    """The fused stage as the forward runs it (k_utd3) and the producer/consumer variant k_utd2 (kept as a measured
    alternative) against an fp32 evaluation with stock ops."""
    m = gpu_vsr_f16.model
    N, h, w = shape
    P = m._packed()
    rs = np.random.RandomState(N * 1000 + h * 10 + w + chain)
    a = torch.from_numpy((rs.randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    with torch.no_grad():
        ref, _ = _stage_reference(m, chain, a.float().permute(0, 3, 1, 2))
        got = m._utd2(a, P["utd2"][chain], N, h, w) if kernel == "roles" else m._utd(a, P["utd"][chain], N, h, w)
        got = got.float().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= 3e-3 * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (3, 20, 70), (1, 2, 2), (1, 33, 31), (2, 37, 95)])
@pytest.mark.parametrize("rps", [0, 6])
def test_stage_builds_bit_identical(gpu_vsr_f16, shape, rps):
    """k_utd3 (one wave per SIMD, hand-ordered step; the default) and k_utd (two waves per SIMD) follow the same
    arithmetic order per accumulator: identical bits, also across row segments and on border strips."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    N, h, w = shape
    P = m._packed()
    a = torch.from_numpy((np.random.RandomState(h * 100 + w).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    lib = L.load()
    outs = []
    try:
        for variant in (0, 1):
            L.check(lib.vsr_sr_utd_variant(variant))
            out = torch.empty((N, h, w, 32), dtype=torch.float16, device="cuda")
            L.check(lib.vsr_sr_utd_f16(L.dptr(a, torch.float16), L.dptr(P["utd"][0], torch.uint8), L.dptr(out, torch.float16),
                                       N, h, w, rps or h, 0, 1, L.stream()))
            outs.append(out)
    finally:
        lib.vsr_sr_utd_variant(0)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


def test_stage_row_segments_agree(gpu_vsr_f16):
    """One march over all rows, several row segments (recomputed halo group) and the flat split give bit-identical maps."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    P = m._packed()
    N, h, w = 2, 37, 45
    a = torch.from_numpy((np.random.RandomState(5).randn(N, h, w, 32) * 10).astype(np.float16)).cuda()
    for fn, blob in ((L.load().vsr_sr_utd2_f16, P["utd2"][0]), (None, P["utd"][0])):
        outs = []
        # (negative: the flat split of the one-wave-per-SIMD build -- that many workgroups share the 2 x 2 x 37 strip rows evenly:
        #  shares inside a strip, spanning strips and planes, of one row, more workgroups than rows)
        for rps in ((h, 16, 5, 1, -3, -4, -7, -50, -148, -1000) if fn is None else (h, 16, 5, 1)):
            out = torch.empty((N, h, w, 32), dtype=torch.float16, device="cuda")
            if fn is None:
                L.check(L.load().vsr_sr_utd_f16(L.dptr(a, torch.float16), L.dptr(blob, torch.uint8), L.dptr(out, torch.float16),
                                                N, h, w, rps, 0, 1, L.stream()))
            else:
                L.check(fn(L.dptr(a, torch.float16), L.dptr(blob, torch.uint8), L.dptr(out, torch.float16), N, h, w, rps, 1, L.stream()))
            outs.append(out)
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (3, 20, 70), (1, 2, 2), (1, 1, 31), (1, 33, 31), (2, 37, 95), (5, 48, 64)])
@pytest.mark.parametrize("rps", [0, 6, 1, -3, -7, -64, -1000])
def test_stage_with_fused_uptran_bit_identical(gpu_vsr_f16, shape, rps):
    """vsr_sr_utd_post_f16 (k_utd3<.., POST = 1>: the next group's uptran 1x1 + PReLU applied to every finished output row inside the
    stage's launch) against the two launches it replaces -- vsr_sr_utd_f16, then vsr_sr_chain1x1_f16 on its output: both tensors bit
    for bit, over whole marches, row segments (recomputed halo, 1-row segments) and the flat split (shares spanning strips / planes)."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    N, h, w = shape
    P = m._packed()
    assert 0 in P["utd_post"] and 3 not in P["utd_post"]      # six groups: stage 0 is followed by another stage, stage 3 is not
    a = torch.from_numpy((np.random.RandomState(h * 100 + w + N).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    lib = L.load()
    ref = torch.empty((N, h, w, 32), dtype=torch.float16, device="cuda")
    L.check(lib.vsr_sr_utd_f16(L.dptr(a, torch.float16), L.dptr(P["utd"][0], torch.uint8), L.dptr(ref, torch.float16), N, h, w, rps or h, 0, 1, L.stream()))
    ref_post = m._chain([dict(ins=[(ref.view(N, h * w, 32), P["ut_w"][3], 32 * 4)], prev=None, bias=P["ut_b"][3], slope=P["ut_a"][3])], N, h * w, keep=[True])[0]
    out = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda")
    post = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda")
    L.check(lib.vsr_sr_utd_post_f16(L.dptr(a, torch.float16), L.dptr(P["utd_post"][0], torch.uint8), L.dptr(out, torch.float16), L.dptr(post, torch.float16),
                                    N, h, w, rps or h, 1, L.stream()))
    assert torch.isfinite(post.float()).all()
    assert torch.equal(out, ref)
    assert torch.equal(post.view(N, h * w, 32), ref_post)


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (3, 20, 70), (1, 2, 2), (1, 1, 31), (1, 33, 31), (2, 37, 95), (5, 48, 64), (8, 12, 32)])
@pytest.mark.parametrize("chain", [0, 3])
def test_stage_on_the_32x32x16_mfma(gpu_vsr_f16, shape, chain):
    """k_utd4 (csrc/sr_utd4.hip: the fused stage on v_mfma_f32_32x32x16_f16, the default build) against the fp32 stock-operator
    evaluation of the stage (the bar of k_utd3's test) and against k_utd3 (same products, K summed in another order: fp16 rounding
    of nearly equal fp32 sums -- a last-place difference on a few values); its own launch geometries (whole marches, row segments,
    the flat split) agree bit for bit; the fused uptran output is bit for bit what the chain kernel makes of the stage's output."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    N, h, w = shape
    P = m._packed()
    rs = np.random.RandomState(N * 1000 + h * 10 + w + chain)
    a = torch.from_numpy((rs.randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    lib = L.load()

    def run(rps, post):
        out = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda")
        op = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda") if post else None
        L.check(lib.vsr_sr_utd4_f16(L.dptr(a, torch.float16), L.dptr(P["utd4"][chain], torch.uint8), L.dptr(out, torch.float16), L.optr(op, torch.float16),
                                    N, h, w, rps, 1, L.stream()), "sr_utd4_f16")
        return out, op
    with torch.no_grad():
        ref, _ = _stage_reference(m, chain, a.float().permute(0, 3, 1, 2))
        got, _ = run(h, False)
        old = m._utd(a, P["utd"][chain], N, h, w)
    assert torch.isfinite(got.float()).all()
    err = (got.float().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err <= 3e-3 * ref.abs().max().item(), (err, ref.abs().max().item())
    d = (got.float() - old.float()).abs()
    assert d.max().item() <= 2e-3 * old.float().abs().max().item() and (d > 0).float().mean().item() < 0.2, (d.max().item(), (d > 0).float().mean().item())
    for rps in (6, 1, -3, -7, -64, -1000):
        o, _ = run(rps, False)
        assert torch.equal(o, got), rps
    if chain == 0:   # the blob of stage 0 carries the next group's uptran slice
        for rps in (h, 6, -7, -64):
            o, op = run(rps, True)
            assert torch.equal(o, got), rps
            want = m._chain([dict(ins=[(got.view(N, h * w, 32), P["ut_w"][3], 32 * 4)], prev=None, bias=P["ut_b"][3], slope=P["ut_a"][3])], N, h * w, keep=[True])[0]
            assert torch.equal(op.view(N, h * w, 32), want), rps


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 95)])
def test_forward_with_fused_uptran_bit_identical(gpu_vsr_f16, shape):
    """The whole SR forward with the uptran slice fused into the first stage of every step (default) and as its own launch."""
    import copy
    m = copy.deepcopy(gpu_vsr_f16.model)
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 7 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    assert m.fuse_uptran
    with torch.no_grad():
        fused = m(x).clone()
        m.fuse_uptran = False
        apart = m(x).clone()
        # ... and with a slope > 1 in the fused 1x1 (the select build k_utd3<false, .., 1>)
        m.block.uptranBlocks[3][1].weight.fill_(1.5)
        apart2 = m(x).clone()
        m.fuse_uptran = True
        fused2 = m(x).clone()
    assert torch.equal(fused, apart)
    assert not m._packed()["post_slopes_le_one"] and torch.equal(fused2, apart2) and not torch.equal(fused2, fused)


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (1, 17, 64)])
def test_deconv_only_mode(gpu_vsr_f16, shape):
    m = gpu_vsr_f16.model
    N, h, w = shape
    P = m._packed()
    a = torch.from_numpy((np.random.RandomState(h).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    with torch.no_grad():
        ref = F.prelu(F.conv_transpose2d(a.float().permute(0, 3, 1, 2), m.out[0].weight, m.out[0].bias, stride=4, padding=2),
                      m.out[1].weight)
        got = m._utd(a, P["utd_out"], N, h, w, deconv_only=True).float().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 1.5e-3 * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("tag", ["16x16", "12x20"])
def test_sr_fp16_matches_reference_golden(golden, gpu_vsr_f16, tag):
    g = golden(f"g1_sr_{tag}")
    taps = {}
    out = gpu_vsr_f16.model(torch.from_numpy(g["x"]).cuda(), taps=taps).cpu().numpy()

    def rel(a, ref):
        a = a.cpu().numpy() if torch.is_tensor(a) else a
        return np.abs(a - ref).max() / np.abs(ref).max()

    assert rel(taps["feat_in"], g["feat_in"]) < 1.5e-3
    for s in range(3):
        assert rel(taps[f"block{s}"], g[f"block{s}"]) < 1e-2, s
    assert rel(taps["prefc2"], g["prefc2"]) < 2e-3
    assert rel(out, g["out"]) < 2e-3
    mse = float(np.mean((out - g["out"]) ** 2))
    assert 10 * np.log10(255.0 ** 2 / mse) > 55.0


def test_sr_fp16_vs_fp32_path_full_width_strip(gpu_vsr, gpu_vsr_f16):
    """Headline width (LR 16x960): both device paths agree to PSNR > 55 dB; the fp16 path is deterministic."""
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (8, 3, 16, 960)).astype(np.float32)).cuda()
    a = gpu_vsr_f16.model(x)
    b = gpu_vsr_f16.model(x)
    assert torch.equal(a, b)
    ref = gpu_vsr.model(x)
    mse = ((a - ref) ** 2).mean().item()
    assert 10 * np.log10(255.0 ** 2 / mse) > 55.0


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33)])
def test_decimated_output_is_a_subset_of_the_full_frame(gpu_vsr_f16, shape):
    """decimate=True (the pixels a nearest x1/4 resize reads: pass 1 of VSR.forward) returns exactly the values of the
    full frame at (4i, 4j)."""
    m = gpu_vsr_f16.model
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    with torch.no_grad():
        full = m(x)
        dec = m(x, decimate=True)
    assert dec.shape == (1, 3, h, w)
    assert torch.equal(dec, full[..., ::4, ::4])


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33), (2, 2)])
@pytest.mark.parametrize("decimate", [False, True])
def test_tail_builds_agree(gpu_vsr_f16, shape, decimate):
    """k_tail3 + fusion MLP with the skip (one wave per SIMD, registers, 3x3 turned around; the forward's pair) against
    k_tail + plain fusion MLP (LDS ring, skip inside the tail): same operands and roundings, only the order of the fp32
    sums differs."""
    m = gpu_vsr_f16.model
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 7 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    try:
        with torch.no_grad():
            m.tail_build = 1
            ref = m(x, decimate=decimate)
            m.tail_build = 3
            got = m(x, decimate=decimate)
    finally:
        m.tail_build = 3
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33), (2, 2), (1, 7)])
def test_chain_builds_bit_identical(gpu_vsr_f16, shape):
    """The 1x1 glue through the streaming builds (specialised on the launch shape, operands requested one tile ahead,
    weights in LDS) against the generic chain kernel: the same MFMAs in the same order, so the frames are equal bit for bit."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 11 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    lib = L.load()
    try:
        with torch.no_grad():
            lib.vsr_sr_chain_variant(1)
            ref = m(x).clone()
            lib.vsr_sr_chain_variant(0)
            got = m(x)
    finally:
        lib.vsr_sr_chain_variant(0)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33), (2, 2), (1, 7)])
def test_fusion_builds_bit_identical(gpu_vsr_f16, shape):
    """Fusion MLP + skip over the raw planes: four pixels per thread on packed fp32 pairs (full frames) against one
    pixel per thread (the decimated pass's build): every product-sum is an explicit fma in both, so they agree exactly."""
    from video_super_resolution_amd import _lib as L
    m = gpu_vsr_f16.model
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 13 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    lib = L.load()
    try:
        with torch.no_grad():
            lib.vsr_sr_chain_variant(2)
            ref = m(x).clone()
            lib.vsr_sr_chain_variant(0)
            got = m(x)
    finally:
        lib.vsr_sr_chain_variant(0)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33), (2, 2), (1, 7)])
@pytest.mark.parametrize("decimate", [False, True])
def test_tail_with_folded_compress_out(gpu_vsr_f16, shape, decimate):
    """The FeedbackBlock's last compress_out applied inside k_tail3's LR load path (k_tail3<.., FOLD>) against its own
    chain launch followed by k_tail3: the 1x1 is computed with the same operands in the same order; only the deconv's K
    index is permuted (fp32 summation order), so the frames agree to 1e-4 of range."""
    m = gpu_vsr_f16.model
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 17 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    try:
        with torch.no_grad():
            m.fold_tail = False
            ref = m(x, decimate=decimate).clone()
            m.fold_tail = True
            got = m(x, decimate=decimate)
    finally:
        m.fold_tail = True
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(12, 20), (9, 40)])
def test_prelu_slopes_above_one_and_negative(gpu_vsr_f16, shape):
    """A trained checkpoint may hold PReLU slopes > 1 or < 0; then max(v, a v) is not PReLU and the kernels switch to
    their select builds (k_utd3<false,..>, k_tail3<false,..>, k_utd<*,false>, prelu_h2's other branch), which the
    synthetic weights (slopes in (0.1, 0.3)) never reach.  One slope of every kind is moved out of (0, 1];
    the fp16 path is compared with the oracle on the same weights at the golden-vector bars."""
    import copy
    from oracle import vsr_oracle as O
    m = copy.deepcopy(gpu_vsr_f16.model)
    b = m.block
    with torch.no_grad():
        b.upBlocks[1][1].weight.fill_(1.25)       # live chain lr0 -> hr1
        b.downtranBlocks[1][1].weight.fill_(-0.15)
        b.downBlocks[2][1].weight.fill_(1.1)
        b.upBlocks[4][1].weight.fill_(-0.1)       # second live chain
        b.uptranBlocks[0][1].weight.fill_(1.5)
        b.compress_out[1].weight.fill_(1.2)
        m.out[1].weight.fill_(1.3)
        b.upBlocks[0][1].weight.fill_(1.4)        # input-independent branch (constant map, exact fp32 kernels)
    assert not m._packed()["slopes_le_one"]
    h, w = shape
    x = torch.from_numpy(np.random.RandomState(h * 31 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32))
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = O.sr_forward(P, x).numpy()
    for decimate in (False, True):
        got = m(x.cuda(), decimate=decimate).cpu().numpy()
        want = ref[..., ::4, ::4] if decimate else ref
        assert np.isfinite(got).all()
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < 2e-3, (decimate, err)
    m.precision = "fp32"
    got32 = m(x.cuda()).cpu().numpy()
    assert np.abs(got32 - ref).max() <= 2e-5 * np.abs(ref).max()
    # the LDS-ring cross-check builds take the same switch
    m.precision = "fp16"
    m.tail_build = 1
    got1 = m(x.cuda()).cpu().numpy()
    assert np.abs(got1 - ref).max() / np.abs(ref).max() < 2e-3


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 95)])
@pytest.mark.parametrize("decimate_first", [True, False])
def test_shared_planes_bit_identical(gpu_vsr_f16, shape, decimate_first):
    """VSR.forward's two SR calls share their first three planes (the LR frames): with `shared`, the second call computes
    head + FeedbackBlock for its other five planes only and must return exactly the frame of a full evaluation; a changed
    weight set or frame size must not reuse stale maps."""
    m = gpu_vsr_f16.model
    h, w = shape
    rs = np.random.RandomState(h * 19 + w)
    a = torch.from_numpy(rs.randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    b = a.clone()
    b[3:] = torch.from_numpy(rs.randint(0, 256, (5, 3, h, w)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref_a, ref_b = m(a, decimate=decimate_first), m(b)
        shared = {"n": 3}
        got_a = m(a, decimate=decimate_first, shared=shared)
        assert shared.get("live") is not None
        got_b = m(b, shared=shared)
        assert torch.equal(got_a, ref_a) and torch.equal(got_b, ref_b)
        # another frame size with the same dict: the kept maps do not apply -> full evaluation, refreshed maps
        c = torch.from_numpy(rs.randint(0, 256, (8, 3, h + 1, w)).astype(np.float32)).cuda()
        assert torch.equal(m(c, shared=shared), m(c))


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33)])
@pytest.mark.parametrize("scale", [4, 2])
def test_precomputed_planes_bit_identical(shape, scale):
    """`precompute_shared`: the FeedbackBlock maps of the first three planes evaluated AHEAD of the calls that use them (VSR.forward
    does so on a side stream beside the guidance trunks); both later calls compute their other five planes only and must return
    exactly the frames of full evaluations -- also when the precompute ran on another stream."""
    from video_super_resolution_amd import SRProjectionModule
    from video_super_resolution_amd.weights import fill_module_
    m = fill_module_(SRProjectionModule(upscale_factor=scale).eval(), seed=0, prefix="model.").cuda()
    h, w = shape
    rs = np.random.RandomState(h * 23 + w + scale)
    a = torch.from_numpy(rs.randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    b = a.clone()
    b[3:] = torch.from_numpy(rs.randint(0, 256, (5, 3, h, w)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref_a, ref_b = m(a, decimate=True), m(b)
        for side, with_tail in ((False, False), (True, False), (False, True), (True, True)):
            shared = {"n": 3}
            live = {k: torch.empty((8, h * w, 32), dtype=torch.float16, device="cuda") for k in (3, 6)}
            if with_tail:   # the shared planes' pre-fusion tail output too: both later calls run their tail on five planes
                live["prefc"] = torch.full((8, 3, scale * h, scale * w), float("nan"), dtype=torch.float32, device="cuda")
            first = a[:3].contiguous()
            if side:
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    m.precompute_shared(first, shared, live)
                torch.cuda.current_stream().wait_stream(st)
            else:
                m.precompute_shared(first, shared, live)
            assert shared.get("live") is not None
            assert ("prefc_all" in shared) == with_tail
            assert torch.equal(m(a, decimate=True, shared=shared), ref_a)
            assert torch.equal(m(b, shared=shared), ref_b)
    with pytest.raises(ValueError):
        m.precompute_shared(a[:2].contiguous(), {"n": 3}, live)


@pytest.mark.parametrize("shape", [(16, 16), (9, 40), (37, 33)])
def test_precomputed_rows_bit_identical(shape):
    """SRProjectionModule.precompute_rows + shared["done_last"]: the maps of the LAST plane evaluated ahead of the call (VSR.forward: the
    estimate plane beside the guidance trunks), those of the first three by precompute_shared; the call itself runs planes 3-6 only."""
    from video_super_resolution_amd import SRProjectionModule
    from video_super_resolution_amd.weights import fill_module_
    m = fill_module_(SRProjectionModule().eval(), seed=0, prefix="model.").cuda()
    h, w = shape
    rs = np.random.RandomState(h * 29 + w)
    a = torch.from_numpy(rs.randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    b = a.clone()
    b[3:] = torch.from_numpy(rs.randint(0, 256, (5, 3, h, w)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref_a, ref_b = m(a, decimate=True), m(b)
        shared = {"n": 3}
        live = {k: torch.full((8, h * w, 32), float("nan"), dtype=torch.float16, device="cuda") for k in (3, 6)}
        live["prefc"] = torch.full((8, 3, 4 * h, 4 * w), float("nan"), dtype=torch.float32, device="cuda")
        m.precompute_shared(a[:3].contiguous(), shared, live)
        m.precompute_rows(a[7:8], live, 7)
        shared["done_last"] = 1
        assert torch.equal(m(a, decimate=True, shared=shared), ref_a)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            m.precompute_rows(b[6:8], live, 6)      # two trailing planes, on another stream
        torch.cuda.current_stream().wait_stream(st)
        shared["done_last"] = 2
        assert torch.equal(m(b, shared=shared), ref_b)
        # every plane but the shared ones evaluated ahead: the call is tail + fusion only
        m.precompute_rows(a[3:8], live, 3)
        shared["done_last"] = 5
        assert torch.equal(m(a, decimate=True, shared=shared), ref_a)
    with pytest.raises(ValueError):
        m.precompute_rows(a[6:8], live, 7)


def test_shared_planes_scale2():
    from video_super_resolution_amd import SRProjectionModule
    from video_super_resolution_amd.weights import fill_module_
    m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), seed=0, prefix="model.").cuda()
    rs = np.random.RandomState(8)
    a = torch.from_numpy(rs.randint(0, 256, (8, 3, 20, 24)).astype(np.float32)).cuda()
    b = a.clone()
    b[3:] = torch.from_numpy(rs.randint(0, 256, (5, 3, 20, 24)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref = m(b)
        shared = {"n": 3}
        m(a, decimate=True, shared=shared)
        assert torch.equal(m(b, shared=shared), ref)
