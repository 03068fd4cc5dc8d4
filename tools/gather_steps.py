"""Gather-kernel time against the number of K steps (square kernels k = 1..11 on one shape, split-K off): intercept = fixed cost of a launch, slope = cost of a K step."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
lib = L.load()
lib.vsr_conv2d_tuning(int(os.environ.get('VSR_TUNING', '1')))      # 1: gather path only (k_conv_igemm_d); 8: the LDS-staged build where the gather path is taken
lib.vsr_conv2d_tuning(1000 + int(os.environ.get('VSR_SPLITK_FILL', '1')))   # 1: split-K off
bits_list = [int(b) for b in sys.argv[1:]] or [0]
for N, cin, H, W, cout in ((4, 32, 67, 120, 64), (2, 512, 32, 60, 512), (4, 64, 135, 240, 64)):
    for k in (1, 3, 5, 7, 9, 11):
        if cin >= 512 and k > 5: continue
        x = igemm.to_nhwc_half(torch.randn(N, cin, H, W, device="cuda"))
        conv = igemm.HConv(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, torch.zeros(cout, device="cuda"), pad=(k - 1) // 2)
        row = []
        for bits in bits_list:
            for _ in range(3): conv(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): conv(x)
            e1.record(); torch.cuda.synchronize()
            row.append(f"{bits}:{e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
        steps = (k * k * (cin // 32) + 1) // 2
        print(f"N{N} {H}x{W} c{cin}->{cout} k{k}: {steps:4d} steps  " + "  ".join(row) + " us", flush=True)
