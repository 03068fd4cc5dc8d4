"""Standalone wall time of each trunk and of the SR call at the bench size (HIP events, one process)."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
from video_super_resolution_amd import _lib as L
for _t in os.environ.get('VSR_TUNING', '').split(','):
    if _t: L.load().vsr_conv2d_tuning(int(_t))
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
x8 = torch.cat([fr[:3].permute(0, 3, 1, 2)] * 2 + [fr[:2].permute(0, 3, 1, 2)], 0).contiguous()
stages = {"flow (2 pairs)": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth x4": lambda: hx(fr),
          "depth x1": lambda: hx(fr[:1]), "vos": lambda: m.VOSModule(fr[0], fr[1], ox), "sr": lambda: m.model(x8),
          "sr decimated": lambda: m.model(x8, decimate=True)}
for name, fn in stages.items():
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1)/6:7.3f} ms")
