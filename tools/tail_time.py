"""Device time of the fused tail k_tail3 (HIP events around its launches, L.TIMER) at 8 planes: full frame and decimated pass, with
the FeedbackBlock's last 1x1 folded into its LR load path and without.  usage: tail_time.py [h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
for fold in (True, False):
    m.fold_tail = fold
    for dec in (False, True):
        for _ in range(3): m(x, decimate=dec)
        torch.cuda.synchronize()
        L.TIMER.reset(); L.TIMER.only = {"sr_tail_f16", "sr_tail_dec_f16"}; L.TIMER.enabled = True
        for _ in range(6): m(x, decimate=dec)
        torch.cuda.synchronize()
        L.TIMER.enabled = False
        for k, (n, ms) in L.TIMER.summary().items():
            print(f"fold={fold} decimate={dec}: {k} {n} launches, {ms:.4f} ms avg")
m.fold_tail = True
