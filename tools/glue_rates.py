"""Achieved rates of the NHWC fp16 glue kernels (pool, nearest-upsample + add) on the hourglass's top-level shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm
torch.set_grad_enabled(False)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, H, W, C, ld) in [(4, 540, 960, 64, 64), (4, 540, 960, 64, 256), (4, 540, 960, 128, 128), (4, 270, 480, 128, 256)]:
    a = torch.randn(N, H // 2, W // 2, ld, device="cuda").half()
    b = torch.randn(N, H, W, ld, device="cuda").half()
    ms = t(lambda: igemm.resize_add(a, ld - C, C, (H, W), b, ld - C))
    gb = (N * H * W * C * 2 * 2 + N * (H // 2) * (W // 2) * C * 2) / 1e9
    mp = t(lambda: igemm.pool2x2(b, ld - C, C, 0))
    gp = (N * H * W * C * 2 + N * (H // 2) * (W // 2) * C * 2) / 1e9
    print(f"N{N} {H}x{W} C{C} (slice of {ld}): resize_add {ms*1e3:7.1f} us = {gb/ms:6.2f} TB/s   pool {mp*1e3:7.1f} us = {gp/mp:6.2f} TB/s")
