"""Runs only the fused up->tran->down kernel at the bench size (UTD_N planes, default 8, x 540 x 960) -- target for rocprofv3 --pmc."""
import os, sys, time
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = int(os.environ.get("UTD_N", "8")), int(sys.argv[1]) if len(sys.argv) > 1 else 540, int(sys.argv[2]) if len(sys.argv) > 2 else 960
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
variants = {"k_utd2 (roles)": lambda: m._utd2(a, P["utd2"][0], N, h, w), "k_utd (uniform)": lambda: m._utd(a, P["utd"][0], N, h, w),
            "k_utd3 post (fused uptran)": lambda: m._utd_post(a, P["utd_post"][0], N, h, w),
            "k_utd4 (32x32x16)": lambda: m._utd4(a, P["utd4"][0], N, h, w), "k_utd4 post": lambda: m._utd4(a, P["utd4"][0], N, h, w, post=True)}
if len(sys.argv) > 4:
    variants = {k: v for k, v in variants.items() if sys.argv[4] in k}
for fn in variants.values():
    for _ in range(2): fn()
torch.cuda.synchronize()
res = {k: [] for k in variants}
for r in range(4):  # interleaved rounds on one device
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / reps)
for k, v in res.items():
    ms = sorted(v)[len(v) // 2]
    print(f"{k:18s} {N}x{h}x{w}: {ms:.3f} ms  -> {N*h*w*294912/ms/1e9:.1f} TFLOP/s")
