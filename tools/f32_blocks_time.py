"""Device time of the float32 (de)convolution blocks of the FeedbackBlock (vsr_sr_deconv_f32 / vsr_sr_conv_f32) on the matrix cores
(variant 0, csrc/sr_f32_mfma.hip) against one pixel per thread (variant 1, csrc/sr_f32.hip).  usage: f32_blocks_time.py [N h w scale]"""
import os, sys, ctypes
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import _lib as L
N, h, w, S = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 540, 960, 2)
K = {4: 8, 2: 6, 3: 7}[S]
lib = L.load()
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(N, 32, h, w).astype(np.float32)).cuda()
wp = torch.from_numpy((rs.randn(K, K, 32, 32) / (4.0 * K)).astype(np.float32)).cuda()
b = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
hr = torch.empty((N, 32, S * h, S * w), dtype=torch.float32, device="cuda")
lr = torch.empty((N, 32, h, w), dtype=torch.float32, device="cuda")
flop = 2.0 * N * h * w * 32 * 32 * K * K
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for variant in (0, 1, 0):
    lib.vsr_sr_f32_variant(variant)
    td = t(lambda: L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(hr), N, h, w, S, None, None, L.cf(0.0), L.stream())))
    tc = t(lambda: L.check(lib.vsr_sr_conv_f32(L.dptr(hr), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(lr), N, h, w, S, L.stream())))
    print(f"variant {variant} ({'MFMA' if variant == 0 else 'one pixel per thread'}): deconv {td:.3f} ms = {flop / td / 1e9:.1f} TFLOP/s, conv {tc:.3f} ms = {flop / tc / 1e9:.1f} TFLOP/s  (of 157)")
lib.vsr_sr_f32_variant(0)
