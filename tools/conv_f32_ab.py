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
    (2, 3, 540, 960, 64, 3, 1), (2, 64, 540, 960, 64, 3, 1), (2, 128, 270, 480, 128, 3, 1), (2, 512, 68, 120, 512, 3, 1),
    (4, 64, 540, 960, 16, 7, 1), (4, 64, 270, 480, 32, 11, 1), (4, 64, 270, 480, 32, 7, 1), (4, 32, 270, 480, 32, 5, 1), (4, 32, 270, 480, 32, 3, 1),
    (4, 32, 135, 240, 64, 7, 1), (2, 256, 135, 240, 256, 3, 1), (1, 64, 540, 960, 16, 11, 1)]
tot = [0.0, 0.0, 0.0]
for N, C, H, W, Co, k, s in cases:
    m = trunk_f32.Conv2dF32(C, Co, k, s, (k - 1) // 2).cuda()
    x = torch.randn(N, C, H, W, device="cuda")
    wp = trunk_f32._pack(m.weight.detach().contiguous())
    b = m.bias.detach()
    pad = (k - 1) // 2
    out = torch.empty((N, Co, (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1), device="cuda")
    flat = t(lambda: trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, k, k, s, pad, pad, 1, out=out))
    sp = None
    if s == 1 and k >= 3 and C > 4:
        try:
            sp = t(lambda: trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, k, k, s, pad, pad, 2, out=out))
        except RuntimeError:
            sp = None
    stock = t(lambda: F.conv2d(x, m.weight, m.bias, stride=s, padding=pad))
    fl = 2.0 * N * (H // s) * (W // s) * C * Co * k * k
    route = trunk_f32._route(N, C, H, W, Co, k, k, s, pad, pad)
    tot[0] += flat; tot[1] += stock; tot[2] += (sp, flat, stock)[(2, 1, 0).index(route)] if (route != 2 or sp is not None) else flat
    sps = f"{sp:8.3f} ms ({fl / sp / 1e9:6.1f} TFLOP/s)" if sp is not None else "       -                  "
    print(f"N{N} {H}x{W} c{C}->{Co} k{k} s{s}: spatial {sps}   flat {flat:8.3f} ms ({fl / flat / 1e9:6.1f} TFLOP/s)   "
          f"stock {stock:8.3f} ms ({fl / stock / 1e9:6.1f} TFLOP/s)   routed: {('stock', 'flat', 'spatial')[route]}", flush=True)
print(f"sum: flat {tot[0]:.2f} ms, stock {tot[1]:.2f} ms, routed {tot[2]:.2f} ms")
