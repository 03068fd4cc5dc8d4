"""Times real trunk layer shapes on the two conv kernels (gather vs LDS patch)."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
cases = [("HG 32->32 k7 4x270x480", 4, 32, 270, 480, 32, 7), ("HG 32->64 k5 4x135x240", 4, 32, 135, 240, 64, 5),
         ("HG 64->64 k11 4x68x120", 4, 64, 68, 120, 64, 11), ("HG 32->32 k3 4x270x480", 4, 32, 270, 480, 32, 3),
         ("VGG 64->64 k3 2x540x960", 2, 64, 540, 960, 64, 3), ("VGG 256->256 k3 2x135x240", 2, 256, 135, 240, 256, 3),
         ("Flow 128->128 k3 2x128x240", 2, 128, 128, 240, 128, 3), ("FlowSD 64->64 k3 2x512x960", 2, 64, 512, 960, 64, 3),
         ("HG final 64->1 k3 3x540x960", 3, 64, 540, 960, 1, 3), ("HG 16-out 64->16 k11 3x270x480", 3, 64, 270, 480, 16, 11),
         ("HG 16-out 32->16 k3 3x270x480", 3, 32, 270, 480, 16, 3), ("predict_flow2 194->2 k3 2x128x240", 2, 194, 128, 240, 2, 3),
         ("predict_flow5 1026->2 k3 2x16x30", 2, 1026, 16, 30, 2, 3), ("side_prep 128->16 k3 2x270x480", 2, 128, 270, 480, 16, 3),
         ("side_prep 512->16 k3 2x68x120", 2, 512, 68, 120, 16, 3), ("fusion inter0 82->16 k3 2x512x960", 2, 82, 512, 960, 16, 3),
         ("fusion pred0 16->2 k3 2x512x960", 2, 16, 512, 960, 2, 3)]
for name, N, cin, H, W, cout, k in cases:
    x = igemm.to_nhwc_half(torch.randn(N, cin, H, W, device="cuda"))
    conv = igemm.HConv(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, torch.zeros(cout, device="cuda"), pad=(k - 1) // 2)
    res = []
    for mode in (1, 2):
        L.load().vsr_conv2d_tuning(mode)
        for _ in range(2): conv(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): conv(x)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{name:42s} gather {res[0]:8.1f} us   patch {res[1]:8.1f} us", flush=True)
L.load().vsr_conv2d_tuning(0)
