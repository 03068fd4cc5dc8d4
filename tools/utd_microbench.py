"""Runs only the fused up->tran->down kernel at the bench size (8 x 540 x 960) -- target for rocprofv3 --pmc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 540, int(sys.argv[2]) if len(sys.argv) > 2 else 960
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
for _ in range(2):
    m._utd(a, P["utd"][0], N, h, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    m._utd(a, P["utd"][0], N, h, w)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"utd {N}x{h}x{w}: {ms:.3f} ms  -> {N*h*w*294912/ms/1e9:.1f} TFLOP/s")
