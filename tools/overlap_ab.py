"""A/B of VSR.overlap_shared at the bench size: ms per recurrent frame with the LR-frame planes' SR maps beside the guidance trunks
(side stream) against the serial order.  usage: overlap_ab.py [h w [scale]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = fill_module_(VSR(upscale_factor=S).eval(), 0).cuda() if S != 4 else fill_module_(VSR().eval(), 0).cuda()
clip = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (14, h, w, 3)).astype(np.float32)).cuda()
for mode, cus in ((False, 256), (True, 96), (True, 128), (True, 160), (True, 192), (True, 256), (False, 256), (True, 96), (True, 128), (True, 160)):
    m.overlap_shared = mode
    m.model.precompute_cus = cus
    est = None
    for t in range(4): est, _ = m(clip[t:t + 3], None, None, est, train=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(4, 12): est, _ = m(clip[t:t + 3], None, None, est, train=False)
    torch.cuda.synchronize()
    print(f"overlap_shared={mode} cus={cus}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per frame")
