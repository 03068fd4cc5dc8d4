"""CPU time to ISSUE whole recurrent forward calls (no synchronise) against their GPU time: how far ahead of the GPU the host runs."""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
clip = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (12, h, w, 3)).astype(np.float32)).cuda()
est = None
for t in range(3):
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
marks = []
for t in range(K):
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
    marks.append(time.perf_counter())
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{K} frames: issue {1e3 * (t1 - t0) / K:.2f} ms per frame, issue + drain {1e3 * (t2 - t0) / K:.2f} ms per frame; host finished {1e3 * (t2 - t1):.1f} ms before the GPU")
print("per-frame issue ms:", " ".join(f"{1e3 * (b - a):.1f}" for a, b in zip([t0] + marks[:-1], marks)))
