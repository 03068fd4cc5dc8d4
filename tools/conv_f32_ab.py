"""Device time of trunk_f32.Conv2dF32 (csrc/conv_f32_nchw.hip) against the stock float32 operator (MIOpen) on layers of the three
trunks at the C2 size.  usage: conv_f32_ab.py"""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from video_super_resolution_amd import trunk_f32
torch.set_grad_enabled(False)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = [  # (N, C, H, W, Co, k, stride)
    (4, 3, 540, 960, 128, 7, 1), (4, 128, 540, 960, 208, 1, 1), (4, 64, 540, 960, 16, 11, 1), (4, 64, 540, 960, 16, 3, 1),
    (4, 128, 270, 480, 128, 1, 1), (4, 32, 270, 480, 32, 7, 1), (4, 64, 135, 240, 64, 11, 1), (4, 32, 67, 120, 64, 7, 1),
    (4, 3, 512, 960, 64, 7, 2), (2, 64, 256, 480, 128, 5, 2), (2, 128, 128, 240, 256, 5, 2), (2, 256, 64, 120, 256, 3, 1),
    (2, 512, 32, 60, 512, 3, 1), (2, 1024, 8, 15, 1024, 3, 1), (2, 1026, 16, 30, 2, 3, 1),
    (2, 3, 540, 960, 64, 3, 1), (2, 64, 540, 960, 64, 3, 1), (2, 128, 270, 480, 128, 3, 1), (2, 512, 68, 120, 512, 3, 1)]
tot = [0.0, 0.0]
for N, C, H, W, Co, k, s in cases:
    m = trunk_f32.Conv2dF32(C, Co, k, s, (k - 1) // 2).cuda()
    x = torch.randn(N, C, H, W, device="cuda")
    own = t(lambda: m(x))
    trunk_f32.ENABLED = False
    try:
        stock = t(lambda: m(x))
    finally:
        trunk_f32.ENABLED = True
    fl = 2.0 * N * (H // s) * (W // s) * C * Co * k * k
    tot[0] += own; tot[1] += stock
    print(f"N{N} {H}x{W} c{C}->{Co} k{k} s{s}: own {own:8.3f} ms ({fl / own / 1e9:6.1f} TFLOP/s)   stock {stock:8.3f} ms ({fl / stock / 1e9:6.1f} TFLOP/s)", flush=True)
print(f"sum: own {tot[0]:.2f} ms, stock {tot[1]:.2f} ms")
