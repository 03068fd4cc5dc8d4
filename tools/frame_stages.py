"""Main-stream time of the four serial stages of VSR.forward at the headline size (guidance 1 incl. the side-stream SR launches it
waits for, SR pass 1, guidance 2, SR pass 2) from events recorded at the stage boundaries (VSR.stage_timing): no synchronisation
inside a frame, steady state, median over frames.  usage: frame_stages.py [frames [lr_h lr_w scale [precision]]]   (environment switches select the routes)"""
import os, sys
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h, w, S = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (540, 960, 4)
m = fill_module_(VSR(upscale_factor=S).eval(), 0).cuda()
if len(sys.argv) > 5: m.precision = m.model.precision = sys.argv[5]
clip = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (n + 6, h, w, 3)).astype(np.float32)).cuda()
est = None
for t in range(4):
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
m.stage_timing = True
rows, frames = [], []
torch.cuda.synchronize()
for t in range(4, 4 + n):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
    frames.append((e0, list(m.stage_events)))
torch.cuda.synchronize()
for e0, ev in frames:
    rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)] + [e0.elapsed_time(ev[4])])
a = np.median(np.array(rows), axis=0)
print(f"guidance 1 {a[0]:6.2f} ms   SR pass 1 {a[1]:6.2f} ms   guidance 2 {a[2]:6.2f} ms   SR pass 2 {a[3]:6.2f} ms   frame {a[4]:6.2f} ms   (median of {n} steady-state frames)")
