"""A/B in one process: trunks with the gather layers through k_conv_igemm_d (default) or k_conv_igemm (tuning mode 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
stages = {"flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth x4": lambda: hx(fr),
          "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}
def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for mode in (11, 0):
        L.load().vsr_conv2d_tuning(mode)
        print(f"mode {mode} ({'128-channel tiles where they pay' if mode == 0 else '64-channel tiles only            '}): " + "  ".join(f"{k} {t(fn):.3f} ms" for k, fn in stages.items()), flush=True)
L.load().vsr_conv2d_tuning(0)
