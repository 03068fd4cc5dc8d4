"""The scale extension (upscale_factor 2 / 3: SRFBN's (kernel, stride) rows (6,2) / (7,3) in place of the reference's
literal (8,4)) on the GPU: exact-fp32 kernels and the fp16/MFMA path (phase deconvolutions + in-place 1x1 + strided conv on
the generic NHWC kernel, csrc/sr_scale.hip for the tail) against

  * the fixtures written by the reference's own forward code with those literals changed (tests/golden/g8_sr_x*.npz),
  * the oracle on ragged sizes, including one large enough for the LDS-patch convolution path (>= 8192 pixels),
  * themselves: decimated == full at (S i, S j); plane chunking bit-identical.

Bars are the x4 bars of test_gpu_sr.py / test_gpu_sr_f16.py: 2e-5 of range (fp32), 2e-3 of range and PSNR > 55 dB (fp16).
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vsr_oracle as O  # noqa: E402
from video_super_resolution_amd import SRProjectionModule  # noqa: E402
from video_super_resolution_amd.weights import fill_module_  # noqa: E402

_cache = {}


def sr_module(scale):
    if scale not in _cache:
        m = fill_module_(SRProjectionModule(upscale_factor=scale).eval(), seed=0, prefix="model.")
        P = {k: v.detach().clone() for k, v in m.state_dict().items()}   # CPU copy for the oracle (before .cuda())
        _cache[scale] = (m.cuda(), P)
    return _cache[scale]


def rel(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("name", ["g8_sr_x2_12x20", "g8_sr_x2_9x7", "g8_sr_x3_6x10"])
def test_scaled_sr_matches_the_scaled_reference_golden(golden, name):
    g = golden(name)
    m, _ = sr_module(int(g["scale"]))
    x = torch.from_numpy(g["x"]).cuda()
    m.precision = "fp32"
    taps = {}
    out = m(x, taps=taps)
    assert rel(out, g["out"]) <= 2e-5 and rel(taps["feat_in"], g["feat_in"]) <= 2e-5
    assert rel(taps["block2"], g["block2"]) <= 2e-5 and rel(taps["prefc2"], g["prefc2"]) <= 2e-5
    m.precision = "fp16"
    taps = {}
    out = m(x, taps=taps).cpu().numpy()
    e = rel(out, g["out"])
    psnr = 10 * np.log10(255.0 ** 2 / float(np.mean((out - g["out"]) ** 2)))
    print(f"[{name} fp16] out {e:.2e} of range, PSNR {psnr:.1f} dB, block2 {rel(taps['block2'], g['block2']):.2e}")
    assert e < 2e-3 and psnr > 55.0 and rel(taps["block2"], g["block2"]) < 1e-2


@pytest.mark.parametrize("scale,hw", [(2, (5, 7)), (2, (2, 2)), (2, (31, 17)), (2, (90, 100)), (3, (9, 33))])
def test_scaled_sr_ragged_sizes_vs_oracle(scale, hw):
    m, P = sr_module(scale)
    x = torch.from_numpy(np.random.RandomState(hw[0] * 100 + hw[1]).randint(0, 256, (8, 3) + hw).astype(np.float32))
    with torch.no_grad():
        ref = O.sr_forward(P, x, upscale_factor=scale).numpy()
    m.precision = "fp32"
    assert rel(m(x.cuda()), ref) <= 2e-5
    m.precision = "fp16"
    full = m(x.cuda())
    assert rel(full, ref) < 2e-3
    dec = m(x.cuda(), decimate=True)
    assert dec.shape == (1, 3) + hw and torch.equal(dec, full[..., ::scale, ::scale])


def test_plane_chunking_is_bit_identical(monkeypatch):
    """At 4K -> 8K the HR maps are walked plane by plane (2 GiB per launch); force the chunking at a small size."""
    from video_super_resolution_amd import sr as srmod
    m, _ = sr_module(2)
    m.precision = "fp16"
    x = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (8, 3, 20, 24)).astype(np.float32)).cuda()
    ref = m(x)
    monkeypatch.setattr(srmod, "_planes_per_chunk", lambda N, H, W: 3)
    assert torch.equal(m(x), ref)
    assert torch.equal(m(x, decimate=True), ref[..., ::2, ::2])


@pytest.mark.parametrize("scale,precision,bar", [(2, "fp32", 60.0), (2, "fp16", 55.0), (3, "fp16", 55.0)])
def test_vsr_forward_scaled_vs_oracle(cpu_vsr, scale, precision, bar):
    """The whole forward with the x2 SR net (BASELINE configs C1 / C2 / C3-B / C5 are labelled x2): guidance trunks as in
    the x4 tests, SR net swapped.  Same image-quality bar as the x4 end-to-end tests (discrete guidance planes)."""
    from video_super_resolution_amd import VSR
    m = VSR(upscale_factor=scale).eval()
    sd = {k: v for k, v in cpu_vsr.state_dict().items() if not k.startswith("model.")}
    m.load_state_dict(sd, strict=False)                     # same seeded guidance weights as the x4 model
    fill_module_(m.model, seed=0, prefix="model.")
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    data = torch.from_numpy(np.random.RandomState(21).randint(0, 256, (3, 64, 72, 3)).astype(np.float32))
    with torch.no_grad():
        ref0 = O.vsr_forward(P, data, None, upscale_factor=scale)
        ref1 = O.vsr_forward(P, data, ref0, upscale_factor=scale)
    m = m.cuda()
    m.precision = m.model.precision = precision
    hf = torch.zeros(3, 64 * scale, 72 * scale, 3, device="cuda")
    out0, loss = m(data.cuda(), None, hf, None, train=False)
    out1, _ = m(data.cuda(), None, hf, out0, train=False)
    assert loss is None and out0.shape == (1, 64 * scale, 72 * scale, 3) and torch.equal(hf[1], out1[0])
    for out, ref in ((out0, ref0), (out1, ref1)):
        err = np.abs(out.cpu().numpy() - ref.numpy())
        psnr = 10 * np.log10(255.0 ** 2 / max(float(np.mean(err ** 2)), 1e-20))
        print(f"[x{scale} e2e {precision}] PSNR(255) {psnr:.2f} dB, p99 {np.percentile(err, 99):.4f}")
        assert psnr > bar, psnr


def test_vsr_forward_x2_early_planes_bit_identical(cpu_vsr):
    """VSR.early_planes in the x2 fp16 configuration (C3-B / C5): plane 7 of each pass evaluated beside the guidance trunks;
    three recurrent frames equal those of the order without it, bit for bit."""
    from video_super_resolution_amd import VSR
    m = VSR(upscale_factor=2).eval()
    m.load_state_dict({k: v for k, v in cpu_vsr.state_dict().items() if not k.startswith("model.")}, strict=False)
    fill_module_(m.model, seed=0, prefix="model.")
    m = m.cuda()
    m.precision = m.model.precision = "fp16"
    m.early_scales = (4, 2)   # (off by default for x2: no gain measured there)
    clip = torch.from_numpy(np.random.RandomState(23).randint(0, 256, (5, 66, 70, 3)).astype(np.float32)).cuda()

    def run(level):
        m.early_planes = level
        est, outs = None, []
        for t in range(3):
            est, _ = m(clip[t:t + 3], None, None, est, train=False)
            outs.append(est.clone())
        torch.cuda.synchronize()
        return outs
    ref = run(0)
    for level in (1, 2):
        for a, b in zip(run(level), ref):
            assert torch.equal(a, b), level


# ------------------------------------------------------------------------------------------------------------------
# k_utd_s2 (csrc/sr_utd_s2.hip): the fused x2 stage against an fp32 stock-op evaluation of the same three layers, against
# the unfused launches it replaces, and against itself across row segmentations and plane counts.
def _stage_reference_x2(m, j, a_nchw):
    import torch.nn.functional as F
    b = m.block
    up, dt, dn = b.upBlocks[j + 1], b.downtranBlocks[j + 1], b.downBlocks[j + 2]
    hr = F.prelu(F.conv_transpose2d(a_nchw, up[0].weight, up[0].bias, stride=2, padding=2), up[1].weight)
    c0 = 32 * (j + 2)
    t = F.prelu(F.conv2d(hr, dt[0].weight[:, c0:c0 + 32], dt[0].bias), dt[1].weight)
    return F.prelu(F.conv2d(t, dn[0].weight, dn[0].bias, stride=2, padding=2), dn[1].weight)


@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 9, 40), (3, 20, 70), (1, 2, 2), (1, 33, 31), (8, 12, 30), (1, 1, 61), (2, 47, 3)])
@pytest.mark.parametrize("chain", [0, 3])
@pytest.mark.parametrize("build", [2, 1])
def test_fused_x2_stage(shape, chain, build):
    """build 2: k_utd_s2w (v_mfma_f32_32x32x16_f16, one wave per SIMD, software-pipelined; the default); 1: k_utd_s2."""
    from video_super_resolution_amd import _lib as L
    from video_super_resolution_amd.sr import _UnfusedStage
    m, _ = sr_module(2)
    m.precision = "fp16"
    m.utd_s2_build = build
    N, h, w = shape
    P = m._packed()
    st = P["stage"][chain]
    assert type(st).__name__ == "_FusedStageS2"
    a = torch.from_numpy((np.random.RandomState(N * 1000 + h * 10 + w + chain).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    with torch.no_grad():
        ref = _stage_reference_x2(m, chain, a.float().permute(0, 3, 1, 2))
        got = st(a, m._chain)
        b = m.block
        unf = _UnfusedStage(b.upBlocks[chain + 1], P["dt_w"][chain + 1], 32 * (chain + 2), P["dt_b"][chain + 1], P["dt_a"][chain + 1],
                            b.downBlocks[chain + 2], 2)(a, m._chain)
    rng = ref.abs().max().item()
    assert torch.isfinite(got.float()).all()
    err = (got.float().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err <= 3e-3 * rng, (err, rng)
    assert (got.float() - unf.float()).abs().max().item() <= 4e-3 * rng
    # row segmentations (recomputed halo pairs) are bit-identical
    lib = L.load()
    for rps in (1, 3, 16):
        out = torch.empty_like(got)
        fn = lib.vsr_sr_utd_s2w_f16 if st.wide else lib.vsr_sr_utd_s2_f16
        L.check(fn(L.dptr(a, torch.float16), L.dptr(st.blob, torch.uint8), L.dptr(out, torch.float16), N, h, w, rps, 1, L.stream()))
        assert torch.equal(out, got), rps


def test_fused_and_unfused_x2_networks_agree(golden):
    import copy
    g = golden("g8_sr_x2_12x20")
    m, _ = sr_module(2)
    m.precision = "fp16"
    x = torch.from_numpy(g["x"]).cuda()
    fused = m(x)
    mu = copy.deepcopy(m)
    mu.fused_s2 = False
    mu._pack = None
    unfused = mu(x)
    assert type(mu._packed()["stage"][0]).__name__ == "_UnfusedStage"
    assert rel(fused, g["out"]) < 2e-3 and rel(unfused, g["out"]) < 2e-3
    assert rel(fused, unfused.cpu().numpy()) < 1e-3


@pytest.mark.parametrize("hw", [(20, 24), (33, 61), (7, 95)])
def test_fused_x2_tail_segmentations_and_unfused_build(monkeypatch, hw):
    """k_tail_s2 (deconvolution + conv_out through an LDS ring): every row segmentation gives the same frame bit for bit
    (full and decimated), and the frame agrees with the unfused tail (phase deconvolutions + k_convout_planes)."""
    import copy
    m, _ = sr_module(2)
    m.precision = "fp16"
    x = torch.from_numpy(np.random.RandomState(hw[0] * 3 + hw[1]).randint(0, 256, (8, 3) + hw).astype(np.float32)).cuda()
    ref, ref_dec = m(x), m(x, decimate=True)
    assert "tail_s2" in m._packed() and torch.equal(ref_dec, ref[..., ::2, ::2])
    orig = type(m)._rows_per_segment
    for rows in (1, 4, 9):
        monkeypatch.setattr(type(m), "_rows_per_segment", staticmethod(lambda N, h, w, cus=256, strip=None, rows=rows: rows))
        assert torch.equal(m(x), ref), rows
        assert torch.equal(m(x, decimate=True), ref_dec), rows
    monkeypatch.setattr(type(m), "_rows_per_segment", staticmethod(orig))
    mu = copy.deepcopy(m)
    mu.fused_s2 = False
    mu._pack = None
    unf = mu(x)
    assert "tail_s2" not in mu._packed()
    assert (unf - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 5, 7), (3, 20, 70), (1, 33, 31)])
def test_fused_x2_stage_flat_variant_bit_identical(shape):
    """k_utd_s2's branch-free step (vsr_sr_utd_s2_variant(1): out-of-image pairs computed and zeroed, partial tiles always
    stored, rows that are not output stored out of range) against the step with its uniform branches."""
    from video_super_resolution_amd import _lib as L
    m, _ = sr_module(2)
    m.precision = "fp16"
    m.utd_s2_build = 1   # (k_utd_s2: the variants are its builds)
    N, h, w = shape
    st = m._packed()["stage"][0]
    a = torch.from_numpy((np.random.RandomState(h * 7 + w).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    lib = L.load()
    try:
        lib.vsr_sr_utd_s2_variant(0)
        ref = st(a, m._chain).clone()
        lib.vsr_sr_utd_s2_variant(1)
        got = st(a, m._chain).clone()
        out = torch.empty_like(got)
        L.check(lib.vsr_sr_utd_s2_f16(L.dptr(a, torch.float16), L.dptr(st.blob, torch.uint8), L.dptr(out, torch.float16), N, h, w, 4, 1, L.stream()))
    finally:
        lib.vsr_sr_utd_s2_variant(0)
    assert torch.equal(got, ref) and torch.equal(out, ref)


@pytest.mark.parametrize("shape,rps", [((2, 5, 7), 0), ((1, 9, 40), 3), ((3, 20, 70), 0), ((1, 2, 2), 1), ((1, 33, 31), 16), ((8, 12, 30), 5),
                                       ((1, 1, 61), 0), ((2, 47, 3), 1), ((5, 64, 90), 7)])
@pytest.mark.parametrize("slope", [None, 1.5])
def test_fused_x2_stage_with_fused_uptran_bit_identical(shape, rps, slope):
    """vsr_sr_utd_s2_post_f16 (k_utd_s2<.., POST>: the next group's uptran 1x1 + PReLU applied to every finished output row inside the x2
    stage's launch) against the two launches it replaces -- vsr_sr_utd_s2_f16, then vsr_sr_chain1x1_f16 on its output: both tensors bit for
    bit, over whole marches and row segments (1-row segments: every row is a segment's last row, reduced after the loop)."""
    from video_super_resolution_amd import _lib as L
    m, _ = sr_module(2)
    m.precision = "fp16"
    N, h, w = shape
    if slope is not None:   # a slope > 1 in the fused 1x1: the select build (min instead of max)
        with torch.no_grad():
            m.block.uptranBlocks[3][1].weight.fill_(slope)
    P = m._packed()
    st = P["stage"][0]
    assert st.has_post and not P["stage"][3].has_post      # six groups: stage 0 is followed by another stage, stage 3 is not
    assert st.post_slopes_le_one == (slope is None)
    le1 = int(st.post_slopes_le_one)
    a = torch.from_numpy((np.random.RandomState(h * 100 + w + N).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
    lib = L.load()
    ref = torch.empty((N, h, w, 32), dtype=torch.float16, device="cuda")
    L.check(lib.vsr_sr_utd_s2_f16(L.dptr(a, torch.float16), L.dptr(st.blob, torch.uint8), L.dptr(ref, torch.float16), N, h, w, rps or h, 1, L.stream()))
    ref_post = m._chain([dict(ins=[(ref.view(N, h * w, 32), P["ut_w"][3], 32 * 4)], prev=None, bias=P["ut_b"][3], slope=P["ut_a"][3])], N, h * w, keep=[True])[0]
    out = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda")
    post = torch.full((N, h, w, 32), float("nan"), dtype=torch.float16, device="cuda")
    L.check(lib.vsr_sr_utd_s2_post_f16(L.dptr(a, torch.float16), L.dptr(st.blob, torch.uint8), L.dptr(out, torch.float16), L.dptr(post, torch.float16),
                                       N, h, w, rps or h, le1, L.stream()))
    assert torch.isfinite(post.float()).all() and (post < 0).any()
    assert torch.equal(out, ref)
    assert torch.equal(post.view(N, h * w, 32), ref_post)
    o2, p2 = st(a, m._chain, post=True)      # (the host wrapper: its own segmentation)
    assert torch.equal(o2, ref) and torch.equal(p2.view(N, h * w, 32), ref_post)


@pytest.mark.parametrize("hw", [(12, 20), (37, 45)])
def test_x2_forward_with_fused_uptran_bit_identical(hw):
    """The whole x2 SR forward with the uptran slice fused into the first stage of every step (default) and as its own launch; also with
    a slope > 1 in the fused 1x1 (the select build)."""
    m, _ = sr_module(2)
    m.precision = "fp16"
    h, w = hw
    x = torch.from_numpy(np.random.RandomState(h * 7 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    assert m.fuse_uptran
    with torch.no_grad():
        fused = m(x).clone()
        m.fuse_uptran = False
        apart = m(x).clone()
        m.block.uptranBlocks[3][1].weight.fill_(1.5)
        apart2 = m(x).clone()
        m.fuse_uptran = True
        fused2 = m(x).clone()
    assert torch.equal(fused, apart)
    assert not m._packed()["stage"][0].post_slopes_le_one and torch.equal(fused2, apart2)


@pytest.mark.parametrize("hw", [(16, 16), (9, 40), (37, 33), (2, 2), (1, 7), (40, 64)])
@pytest.mark.parametrize("decimate", [False, True])
def test_x2_tail_with_folded_compress_out_bit_identical(hw, decimate):
    """The FeedbackBlock's last compress_out applied inside k_tail_s2's LR load path (k_tail_s2<.., FOLD>, vsr_sr_tail_s2_fold_f16) against
    its own chain launch followed by k_tail_s2: the 1x1 with the same operands in the same order, its output written to the ring in natural
    channel order -- the frames are equal bit for bit (full frames and the decimated pass-1 output; with shared planes: the VSR tests)."""
    m, _ = sr_module(2)
    m.precision = "fp16"
    h, w = hw
    x = torch.from_numpy(np.random.RandomState(h * 17 + w).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
    m.fold_tail = True
    assert m._packed()["tail_s2_fold"]
    with torch.no_grad():
        got = m(x, decimate=decimate).clone()
        m.fold_tail = False
        ref = m(x, decimate=decimate).clone()
        # ... and with a slope > 1 in compress_out (min instead of max)
        m.block.compress_out[1].weight.fill_(1.25)
        ref2 = m(x, decimate=decimate).clone()
        m.fold_tail = True
        got2 = m(x, decimate=decimate).clone()
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)
    assert torch.equal(got2, ref2)
