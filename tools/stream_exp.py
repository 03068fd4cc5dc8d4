"""Time of the 1x1 streaming kernel on the hourglass's fused inception reductions."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = [(4, 128, 540, 960, 208), (4, 128, 270, 480, 224), (4, 128, 270, 480, 128)]
if len(sys.argv) > 1: cases = cases[:1]
for (N, cin, H, W, cout) in cases:
    x = torch.randn(N, H, W, cin, device="cuda").half()
    w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    conv = igemm.HConv(w, torch.zeros(cout, device="cuda"), stride=1, pad=0, act=igemm.ACT_RELU)
    r = []
    for mode, nm in ((0, "stream"),) if len(sys.argv) > 1 else ((0, "stream"), (1, "gather")):
        L.load().vsr_conv2d_tuning(mode)
        ms = t(lambda: conv(x))
        r.append(f"{nm} {ms*1e3:7.1f} us")
    L.load().vsr_conv2d_tuning(0)
    gb = N * H * W * (cin + ((cout + 15) // 16) * 16) * 2 / 1e9
    print(f"N{N} {H}x{W} c{cin}->{cout} 1x1 ({gb:.2f} GB): " + " | ".join(r))
