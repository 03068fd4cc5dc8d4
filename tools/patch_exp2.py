"""EXPERIMENT (timing only, wrong results): k_conv_patch with parts removed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, cin, H, W, cout, k) in [(2, 32, 512, 960, 64, 3), (2, 64, 256, 480, 128, 3), (2, 128, 128, 240, 128, 3)]:
    x = torch.randn(N, H, W, cin, device="cuda").half()
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    conv = igemm.HConv(w, torch.zeros(cout, device="cuda"), stride=1, pad=k // 2, act=igemm.ACT_RELU)
    r = []
    for mode, nm in ((5, "full"), (10, "no stores"), (11, "no staging"), (12, "no MFMA loop")):
        L.load().vsr_conv2d_tuning(mode)
        r.append(f"{nm} {t(lambda: conv(x))*1e3:7.1f} us")
    L.load().vsr_conv2d_tuning(0)
    print(f"N{N} {H}x{W} c{cin}->{cout} k{k}: " + " | ".join(r))
