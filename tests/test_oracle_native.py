"""oracle/native_ops.c (restatement of the three CUDA extensions, which cannot run here and have no reference
fixtures) cross-checked against independent stock PyTorch formulations.  CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import native


def test_resample2d_vs_grid_sample():
    rs = np.random.RandomState(0)
    img = rs.randn(2, 3, 40, 56).astype(np.float32)
    flow = (rs.randn(2, 2, 40, 56) * 8).astype(np.float32)  # plenty of out-of-range targets -> border clamp
    out = native.resample2d(img, flow)
    B, C, H, W = img.shape
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    grid = np.stack([2 * (xs[None] + flow[:, 0]) / (W - 1) - 1, 2 * (ys[None] + flow[:, 1]) / (H - 1) - 1], -1)
    ref = F.grid_sample(torch.from_numpy(img), torch.from_numpy(grid.astype(np.float32)), mode="bilinear",
                        padding_mode="border", align_corners=True).numpy()
    assert np.abs(out - ref).max() < 1e-4


def test_resample2d_integer_flow_is_a_shift_and_nearest_mode():
    rs = np.random.RandomState(1)
    img = rs.randn(1, 2, 9, 11).astype(np.float32)
    flow = np.zeros((1, 2, 9, 11), np.float32)
    flow[:, 0] = 2
    flow[:, 1] = -1
    out = native.resample2d(img, flow)
    ref = img[:, :, np.clip(np.arange(9) - 1, 0, 8)][:, :, :, np.clip(np.arange(11) + 2, 0, 10)]
    np.testing.assert_array_equal(out, ref)
    flow[:, 0] = 1.6
    np.testing.assert_array_equal(native.resample2d(img, flow, bilinear=False),
                                  img[:, :, np.clip(np.arange(9) - 1, 0, 8)][:, :, :, np.clip(np.arange(11) + 2, 0, 10)])


def test_channelnorm():
    x = np.random.RandomState(2).randn(2, 3, 17, 19).astype(np.float32)
    np.testing.assert_allclose(native.channelnorm(x), np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True)), rtol=1e-6)


def test_correlation_vs_shift_multiply_mean():
    rs = np.random.RandomState(3)
    f1 = rs.randn(1, 64, 12, 14).astype(np.float32)
    f2 = rs.randn(1, 64, 12, 14).astype(np.float32)
    out = native.correlation(f1, f2, 20, 1, 20, 1, 2)
    assert out.shape == (1, 441, 12, 14)
    p2 = np.pad(f2, ((0, 0), (0, 0), (20, 20), (20, 20)))
    for tj in (-10, -3, 0, 7, 10):
        for ti in (-10, 0, 1, 10):
            ref = (f1 * p2[:, :, 20 + 2 * tj:32 + 2 * tj, 20 + 2 * ti:34 + 2 * ti]).mean(1)
            np.testing.assert_allclose(out[:, (tj + 10) * 21 + ti + 10], ref, atol=1e-6)
