"""Single layers through the patch builds (k_conv_patch_r8 | k_conv_patch / rows | gather), one process."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
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
for (N, cin, H, W, cout, k) in [(2, 32, 512, 960, 64, 3), (2, 64, 256, 480, 128, 3), (2, 128, 128, 240, 128, 3), (4, 64, 270, 480, 32, 7), (4, 32, 270, 480, 32, 7), (4, 64, 135, 240, 64, 11), (4, 64, 540, 960, 16, 11), (4, 64, 540, 960, 16, 3), (4, 32, 270, 480, 32, 5), (4, 32, 67, 120, 64, 7), (2, 96, 512, 960, 16, 3)]:
    x = torch.randn(N, H, W, cin, device="cuda").half()
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    conv = igemm.HConv(w, torch.zeros(cout, device="cuda"), stride=1, pad=k // 2, act=igemm.ACT_RELU)
    fl = 2.0 * N * H * W * cin * cout * k * k
    r = []
    for mode in (0, 5, 1):
        L.load().vsr_conv2d_tuning(mode)
        ms = t(lambda: conv(x))
        r.append(f"{['r8', 'patch', 'gather'][(0, 5, 1).index(mode)]} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF")
    L.load().vsr_conv2d_tuning(0)
    print(f"N{N} {H}x{W} c{cin}->{cout} k{k}: " + " | ".join(r))
