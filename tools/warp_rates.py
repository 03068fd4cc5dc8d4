"""Time of the flow upsample + warp + concat launch (vsr_flownet_up_warp_concat16_f16) at the benchmark size, both builds
(0: LDS-staged tile + DPP neighbour hand-over, 1: thread-per-pixel gathers), rounds interleaved, on a smooth flow (the
FlowNet case) and on a rough one."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import _lib as L
lib = L.load()
B, H, W = 2, 512, 960
x = torch.rand(B, 6, H, W, device="cuda")
out = torch.empty((B, H, W, 16), dtype=torch.float16, device="cuda")
def t(reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.vsr_flownet_up_warp_concat16_f16(L.dptr(x), L.dptr(f2, torch.float16), 32, 1, L.cf(20.0), L.cf(0.05), L.dptr(out, torch.float16), B, H, W, L.stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, sig in (("smooth (sigma 0.1 px at 1/4 res)", 0.005), ("~2 px", 0.1), ("~20 px", 1.0), ("~160 px (fallback)", 8.0)):
    f2 = torch.zeros((B, H // 4, W // 4, 32), dtype=torch.float16, device="cuda")
    f2[..., :2] = (torch.randn(B, H // 4, W // 4, 2, device="cuda") * sig).half()
    r = {0: [], 1: []}
    for rnd in range(4):
        for v in (0, 1):
            lib.vsr_flownet_warp_variant(v); t(5); r[v].append(t())
    lib.vsr_flownet_warp_variant(1)
    print(f"{name:36s} LDS-staged {min(r[0]):6.1f} us   gathers {min(r[1]):6.1f} us")
