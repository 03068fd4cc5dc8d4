"""k4 s2 transposed convolutions of FlowNet2 at the headline size: heuristic (patch build for few out-channels) vs gather only."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import igemm, _lib as L
torch.set_grad_enabled(False)
cases = [(2, 192, 256, 480, 16), (2, 128, 128, 240, 32), (2, 32, 256, 480, 2), (2, 32, 128, 240, 2), (2, 416, 64, 120, 64), (2, 32, 64, 120, 2)]
lib = L.load()
for N, cin, H, W, cout in cases:
    x = igemm.to_nhwc_half(torch.randn(N, cin, H, W, device="cuda"))
    dc = igemm.HDeconv4s2(torch.randn(cin, cout, 4, 4, device="cuda") / (cin * 4) ** 0.5, torch.zeros(cout, device="cuda"), act=igemm.ACT_LEAKY)
    row = []
    for mode in (0, 1):
        lib.vsr_conv2d_tuning(mode)
        for _ in range(3): dc(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): dc(x)
        e1.record(); torch.cuda.synchronize()
        row.append(f"mode {mode}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
    lib.vsr_conv2d_tuning(0)
    print(f"deconv4s2 N{N} {H}x{W} c{cin}->{cout}: " + "   ".join(row) + " us", flush=True)
