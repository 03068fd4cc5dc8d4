"""cProfile of the host side of steady-state VSR.forward calls (no synchronise inside): where the ~11 ms per frame of issue time go."""
import os, sys, cProfile, pstats, io
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
pr = cProfile.Profile()
pr.enable()
for t in range(6):
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
