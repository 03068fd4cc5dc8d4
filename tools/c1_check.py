"""Config C1 with the checker beside it (test infrastructure, NOT part of the package): runs
`video_super_resolution_amd.driver.run_c1` on the GPU, then the CPU oracle on the same windows, and prints the driver's
line with the oracle's frame rate and the PSNR between the two.   python tools/c1_check.py [--lr 64 --frames 5 --scale 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vsr_oracle as O  # noqa: E402
from video_super_resolution_amd import driver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lr", type=int, default=128)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--scale", type=int, default=4, choices=[2, 3, 4])
ap.add_argument("--precision", default="fp32", choices=["fp16", "fp32"])
a = ap.parse_args()
line, model, datas, outs = driver.run_c1(a.lr, a.frames, a.scale, a.precision)
P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
est, mse, T = None, 0.0, outs.shape[0]
t0 = time.perf_counter()
lr_cpu = O.make_lr(datas.cpu(), a.scale)
for t in range(T):
    with torch.no_grad():
        est = O.vsr_forward(P, lr_cpu[t], est, upscale_factor=a.scale)
    mse += float(((outs[t].cpu() - est[0]) ** 2).mean())
line["cpu_oracle_frames_per_s"] = round(T / (time.perf_counter() - t0), 5)
line["psnr_vs_oracle_db"] = round(10 * np.log10(255.0 ** 2 / max(mse / T, 1e-20)), 2)
print(json.dumps(line))
