"""Kernel times of real trunk layer shapes: the gather kernel (LDS-patch kernels switched off) or, with VSR_TUNING=0, whatever the heuristic picks."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
cases = [("HG c32->64 k7 4x67x120", 4, 32, 67, 120, 64, 7, 1), ("HG c32->64 k3 4x67x120", 4, 32, 67, 120, 64, 3, 1),
         ("Flow c512->512 k3 2x32x60", 2, 512, 32, 60, 512, 3, 1), ("Flow c256->256 k3 2x64x120", 2, 256, 64, 120, 256, 3, 1),
         ("Flow c256->512 k3 s2 2x64x120", 2, 256, 64, 120, 512, 3, 2), ("Flow c128->256 k5 s2 2x128x240", 2, 128, 128, 240, 256, 5, 2),
         ("Flow c1024->1024 k3 2x8x15", 2, 1024, 8, 15, 1024, 3, 1), ("Flow c224->2 k3 2x128x240", 2, 224, 128, 240, 2, 3, 1),
         ("FlowSD c32->64 k3 2x512x960", 2, 32, 512, 960, 64, 3, 1), ("VGG c64->64 k3 2x540x960", 2, 64, 540, 960, 64, 3, 1),
         ("VGG c256->256 k3 2x135x240", 2, 256, 135, 240, 256, 3, 1), ("HG c64->32 k7 4x270x480", 4, 64, 270, 480, 32, 7, 1)]   # (patch-kernel shapes: VSR_TUNING=0)
lib = L.load()
if len(sys.argv) > 1: cases = [cases[int(sys.argv[1])]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib.vsr_conv2d_tuning(int(os.environ.get('VSR_TUNING', '1')))   # 1: gather path only; 0: the heuristic (LDS-patch kernels where they pay)
for name, N, cin, H, W, cout, k, st in cases:
    x = igemm.to_nhwc_half(torch.randn(N, cin, H, W, device="cuda"))
    conv = igemm.HConv(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, torch.zeros(cout, device="cuda"), stride=st, pad=(k - 1) // 2)
    for _ in range(3): conv(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): conv(x)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / reps * 1e3:6.1f} us", flush=True)
