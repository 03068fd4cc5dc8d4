"""Trunk wall times against the split-K fill threshold (workgroups below which K is split)."""
import os, sys, time
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
fns = {"flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth4": lambda: hx(fr), "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}
for thr in (64, 96, 128, 160, 192, 256, 96, 128, 192, 256):
    L.load().vsr_conv2d_tuning(1000 + thr)
    row = []
    for name, fn in fns.items():
        fn(); fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        row.append(f"{name} {sorted(ts)[2]:.2f} ms")
    print(f"fill threshold {thr:4d}: " + "  ".join(row), flush=True)
