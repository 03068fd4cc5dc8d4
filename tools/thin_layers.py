"""Thin-output layers (predict_flow: c -> 2, 3x3) under the kernel-selection modes: 0 heuristic, 1 gather only, 2 LDS-patch whenever legal."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
cases = [("c32->2 2x512x960", 2, 32, 512, 960, 2), ("c64->2 2x128x240", 2, 64, 128, 240, 2), ("c128->2 2x128x240", 2, 128, 128, 240, 2),
         ("c224->2 2x128x240", 2, 224, 128, 240, 2), ("c416->2 2x64x120", 2, 416, 64, 120, 2), ("c128->2 2x64x120", 2, 128, 64, 120, 2),
         ("c800->2 2x32x60", 2, 800, 32, 60, 2), ("c256->2 2x32x60", 2, 256, 32, 60, 2), ("c1056->2 2x16x30", 2, 1056, 16, 30, 2),
         ("c64->1 4x540x960", 4, 64, 540, 960, 1), ("c64->16 4x540x960", 4, 64, 540, 960, 16), ("c256->16 2x135x240", 2, 256, 135, 240, 16),
         ("c512->16 2x68x120", 2, 512, 68, 120, 16)]
lib = L.load()
for name, N, cin, H, W, cout in cases:
    x = igemm.to_nhwc_half(torch.randn(N, cin, H, W, device="cuda"))
    conv = igemm.HConv(torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5, torch.zeros(cout, device="cuda"), pad=1)
    row = []
    for mode in (0, 1, 2):
        lib.vsr_conv2d_tuning(mode)
        for _ in range(3): conv(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): conv(x)
        e1.record(); torch.cuda.synchronize()
        row.append(f"mode {mode}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
    lib.vsr_conv2d_tuning(0)
    print(f"{name:22s} " + "   ".join(row) + " us", flush=True)
