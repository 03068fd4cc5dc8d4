"""compress_out folded into k_tail3 against chain launch + k_tail3: agreement and time (one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
for dec in (False, True):
    outs = {}
    for fold in (False, True):
        m.fold_tail = fold
        outs[fold] = m(x, decimate=dec).clone()
    d = (outs[True] - outs[False]).abs()
    print(f"decimate={dec}: max |fold - unfused| = {d.max().item():.4g}, mean {d.mean().item():.3g} of range {outs[False].abs().max().item():.4g}; nan: {torch.isnan(outs[True]).sum().item()}")
for rnd in range(2):
    for fold in (False, True):
        m.fold_tail = fold
        for _ in range(2): m(x); m(x, decimate=True)
        torch.cuda.synchronize()
        L.TIMER.enabled = True; L.TIMER.reset()
        for _ in range(3): m(x); m(x, decimate=True)
        torch.cuda.synchronize(); L.TIMER.enabled = False
        S = L.TIMER.summary()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): m(x); m(x, decimate=True)
        e1.record(); torch.cuda.synchronize()
        print(f"fold={fold}: full + decimated SR call {e0.elapsed_time(e1)/5:.3f} ms   " + "  ".join(f"{k.replace('sr_','')} {n}x{ms:.3f}" for k, (n, ms) in S.items() if "tail" in k or "x1" in k))
m.fold_tail = True
