"""The x2 SR net (fused k_utd_s2 / k_tail_s2 path) at BASELINE.json's full sizes: C3-B (8 planes of LR 1080x1920 -> 2160x3840)
and C5 (LR 2160x3840 -> 4320x7680, one 17 GB feature map had the x2 map been materialised).  As for the x4 headline size
(test_gpu_headline_size.py) the full-size run is pinned by size-independent properties -- the pass-1 frame equals the full frame
at the pixels (2i, 2j) bit for bit; the run is deterministic -- and by a 64x64-LR band of the full frame against the oracle's
frame of the same band computed from a 96x96 crop (16 LR pixels of halo; the x2 net's receptive radius is below that)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vsr_oracle as O  # noqa: E402
from video_super_resolution_amd import SRProjectionModule  # noqa: E402
from video_super_resolution_amd.weights import fill_module_  # noqa: E402

HALO, SIZE = 16, 64


@pytest.fixture(scope="module")
def sr2():
    m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), seed=0, prefix="model.")
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    m.precision = "fp16"
    return m, P


@pytest.mark.parametrize("hw,band", [((1080, 1920), (400, 700)), ((2160, 3840), (1500, 3700))])
def test_x2_full_size_properties_and_band(sr2, hw, band):
    m, P = sr2
    h, w = hw
    top, left = band
    x = torch.from_numpy(np.random.RandomState(h + w).randint(0, 256, (8, 3, h, w)).astype(np.float32))
    crop = x[:, :, top - HALO:top + SIZE + HALO, left - HALO:left + SIZE + HALO].contiguous()
    with torch.no_grad():
        ref = O.sr_forward(P, crop, upscale_factor=2)[..., 2 * HALO:2 * (HALO + SIZE), 2 * HALO:2 * (HALO + SIZE)].numpy()
    xg = x.cuda()
    full = m(xg)
    assert full.shape == (1, 3, 2 * h, 2 * w) and torch.isfinite(full).all()
    dec = m(xg, decimate=True)
    assert torch.equal(dec, full[..., ::2, ::2])
    got = full[..., 2 * top:2 * (top + SIZE), 2 * left:2 * (left + SIZE)].cpu().numpy()
    assert torch.equal(m(xg), full)          # deterministic
    del full, dec
    torch.cuda.empty_cache()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    psnr = 10 * np.log10(255.0 ** 2 / max(float(np.mean((got - ref) ** 2)), 1e-30))
    print(f"[x2 band at {h}x{w}] max err / range = {err:.3e}, PSNR(255) = {psnr:.2f} dB")
    assert err <= 5e-4 and psnr > 90.0, (err, psnr)


def test_c2_fp32_full_size_band_and_mfma_build_identity():
    """BASELINE config C2's geometry in float32: 8 planes of LR 540x960, x2 (the float32 SR blocks on v_mfma_f32_32x32x2_f32 with
    ds_bpermute-shifted operands, csrc/sr_f32_mfma.hip -- VERDICT r3 weak 2: until now run at this size only inside bench.py).
    (i) a 64x64-LR band of the full frame against the oracle's frame of the same band (96x96 crop), at the float32 bar 2e-5 of
    range; (ii) the k6 s2 ConvTranspose2d / Conv2d blocks at 8 x 540 x 960: the MFMA build (default) equals the
    one-pixel-per-thread build bit for bit at the full size too (ragged last column tile: 960 = 15 x 64, 1920 = 30 x 64; 540 rows
    = 135 workgroup rows)."""
    import ctypes
    from video_super_resolution_amd import _lib as L
    h, w = 540, 960
    top, left = 300, 820
    m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), seed=0, prefix="model.")
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    m.precision = "fp32"
    x = torch.from_numpy(np.random.RandomState(h + w + 2).randint(0, 256, (8, 3, h, w)).astype(np.float32))
    crop = x[:, :, top - HALO:top + SIZE + HALO, left - HALO:left + SIZE + HALO].contiguous()
    with torch.no_grad():
        ref = O.sr_forward(P, crop, upscale_factor=2)[..., 2 * HALO:2 * (HALO + SIZE), 2 * HALO:2 * (HALO + SIZE)].numpy()
    with torch.no_grad():
        full = m(x.cuda())
    assert full.shape == (1, 3, 2 * h, 2 * w) and torch.isfinite(full).all()
    got = full[..., 2 * top:2 * (top + SIZE), 2 * left:2 * (left + SIZE)].cpu().numpy()
    del full
    torch.cuda.empty_cache()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"[C2 fp32 band at {h}x{w}] max err / range = {err:.3e}")
    assert err <= 2e-5, err
    # (ii) the two block kernels at the full size, MFMA build against the per-pixel build
    lib = L.load()
    lib.vsr_sr_f32_variant.restype = ctypes.c_int
    rs = np.random.RandomState(7)
    a = torch.from_numpy(rs.randn(8, 32, h, w).astype(np.float32)).cuda()
    wd = torch.from_numpy((rs.randn(6, 6, 32, 32) / 24.0).astype(np.float32)).cuda()   # [ky][kx][in][out]
    wc = torch.from_numpy((rs.randn(6, 6, 32, 32) / 24.0).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
    outs = []
    try:
        for variant in (0, 1):
            lib.vsr_sr_f32_variant(variant)
            hr = torch.empty((8, 32, 2 * h, 2 * w), dtype=torch.float32, device="cuda")
            L.check(lib.vsr_sr_deconv_f32(L.dptr(a), L.dptr(wd), L.dptr(b), L.cf(0.2), L.dptr(hr), 8, h, w, 2, None, None, L.cf(0.0), L.stream()), "deconv")
            lr = torch.empty((8, 32, h, w), dtype=torch.float32, device="cuda")
            L.check(lib.vsr_sr_conv_f32(L.dptr(hr), L.dptr(wc), L.dptr(b), L.cf(0.2), L.dptr(lr), 8, h, w, 2, L.stream()), "conv")
            outs.append((hr, lr))
    finally:
        lib.vsr_sr_f32_variant(0)
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0])
