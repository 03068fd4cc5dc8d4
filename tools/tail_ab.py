"""k_tail3 against k_tail: agreement and time (one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
outs = {}
for dec in (False, True):
    for b in (1, 3):
        m.tail_build = b
        outs[(b, dec)] = m(x, decimate=dec).clone()
    d = (outs[(3, dec)] - outs[(1, dec)]).abs()
    print(f"decimate={dec}: max |k_tail3 - k_tail| = {d.max().item():.4g} of range {outs[(1, dec)].abs().max().item():.4g}; nan: {torch.isnan(outs[(3, dec)]).sum().item()}")
for b in (1, 3):
    m.tail_build = b
    for _ in range(2): m(x)
    torch.cuda.synchronize()
    L.TIMER.enabled = True; L.TIMER.reset()
    for _ in range(5): m(x); m(x, decimate=True)
    torch.cuda.synchronize(); L.TIMER.enabled = False
    S = L.TIMER.summary()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): m(x)
    e1.record(); torch.cuda.synchronize()
    print(f"build {b}: full tail {S['sr_tail_f16'][1]:.3f} ms   decimated tail {S['sr_tail_dec_f16'][1]:.3f} ms   whole SR call {e0.elapsed_time(e1)/5:.3f} ms")
m.tail_build = 3
