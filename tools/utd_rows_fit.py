"""k_utd3's launch time against the rows a workgroup marches: 5 planes x h x 960 for several h (flat split: 256 shares of 5 x 31 x h / 256 rows)
and 8 planes (grid mode: one march of h rows per workgroup, 248 workgroups) -> least-squares line: us per row, fixed us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
m.precision = "fp16"
P = m._packed()
w = 960
for N in (5, 8, 4, 3):
    pts = []
    for h in (68, 135, 270, 405, 540, 810, 1080):
        a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
        rows = m._rows_per_segment(N, h, w, flat_ok=True)
        per = -(-(N * 31 * h) // 256) if rows < 0 else rows
        for _ in range(3):
            m._utd(a, P["utd"][0], N, h, w)
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m._utd(a, P["utd"][0], N, h, w)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        t = sorted(ts)[2]
        pts.append((per, t))
        print(f"N={N} h={h:4d} rows_per_seg={rows:5d} rows/wg={per:4d}  {t:8.1f} us  {N*h*w*294912/t/1e6:7.1f} TFLOP/s  ({t/per:.3f} us/row)")
    x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
    b, a0 = np.polyfit(x, y, 1)
    print(f"N={N}: {b:.4f} us per row + {a0:.1f} us per launch")
