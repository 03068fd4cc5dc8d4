"""A/B in one process: hourglass with its two arms per level on separate streams or on one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
hx = m._depth_exec.get()
def t(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
outs = {}
for rnd in range(2):
    for conc in (False, True):
        hx.concurrent = conc
        outs[conc] = hx(fr).clone()
        print(f"concurrent={conc}: depth x4 {t(lambda: hx(fr)):.3f} ms   x1 {t(lambda: hx(fr[:1])):.3f} ms", flush=True)
hx.concurrent = True
print("equal:", torch.equal(outs[False], outs[True]))
