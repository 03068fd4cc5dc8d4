"""Runs ONE guidance trunk a few times (target for rocprofv3 --kernel-trace --stats): usage probe_one_trunk.py flow|depth|depth1|vos|sr [reps]"""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
from video_super_resolution_amd import _lib as L
for _t in os.environ.get('VSR_TUNING', '').split(','):
    if _t: L.load().vsr_conv2d_tuning(int(_t))
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h, w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (540, 960)
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
x8 = torch.cat([fr[:3].permute(0, 3, 1, 2)] * 2 + [fr[:2].permute(0, 3, 1, 2)], 0).contiguous()
fn = {"sr": lambda: m.model(x8), "flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx),
      "depth": lambda: hx(fr), "depth1": lambda: hx(fr[:1]), "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}[which]
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1)/reps:.2f} ms per call")
if os.environ.get("VSR_ROUTES_OUT"):   # the (layer label, kernel route) list of one more call, in launch order
    L.ROUTES.calls = []; L.ROUTES.enabled = True; fn(); torch.cuda.synchronize(); L.ROUTES.enabled = False
    with open(os.environ["VSR_ROUTES_OUT"], "w") as f:
        for lab, route in L.ROUTES.calls: f.write(f"{lab}\t{route}\n")
