"""Time of the fused tail kernel alone at the bench size (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, 540, 960
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (N, 3, h, w)).astype(np.float32)).cuda()
for _ in range(2): m(x)
torch.cuda.synchronize()
L.TIMER.enabled = True; L.TIMER.reset()
for _ in range(5): m(x)
torch.cuda.synchronize()
L.TIMER.enabled = False
for k, (n_, ms) in L.TIMER.summary().items():
    print(f"{k:24s} x{n_:3d}  {ms:.4f} ms")
