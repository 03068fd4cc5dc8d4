"""Device time of the fused x2 stage k_utd_s2 (8 planes, LR 1080x1920 by default) inside SRProjectionModule(upscale_factor=2):
HIP events around its launches (L.TIMER), after a warm-up.  usage: s2_time.py [h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
for _ in range(3): m(x)
torch.cuda.synchronize()
L.TIMER.reset(); L.TIMER.only = {"sr_utd_s2_f16"}; L.TIMER.enabled = True
for _ in range(4): m(x)
torch.cuda.synchronize()
L.TIMER.enabled = False
for k, (n, ms) in L.TIMER.summary().items():
    flop = 8 * h * w * 155648.0
    print(f"{k}: {n} launches, {ms:.4f} ms avg = {flop / (ms * 1e-3) / 1e12:.1f} TFLOP/s = {flop / (ms * 1e-3) / 2.5e15:.3f} of peak")
