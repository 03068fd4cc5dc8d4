"""CPU time to ISSUE each guidance trunk (no synchronise) vs its GPU wall time: is the guidance launch-bound?"""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
x8 = torch.cat([fr[:3].permute(0, 3, 1, 2)] * 2 + [fr[:2].permute(0, 3, 1, 2)], 0).contiguous()
fns = {"flow (2 pairs)": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx),
       "depth x4": lambda: hx(fr), "depth x1": lambda: hx(fr[:1]), "vos": lambda: m.VOSModule(fr[0], fr[1], ox), "SR call": lambda: m.model(x8)}
for name, fn in fns.items():
    fn(); fn(); torch.cuda.synchronize()
    iss, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        iss.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
    print(f"{name:16s} issue {sorted(iss)[2]:6.2f} ms   issue+drain {sorted(tot)[2]:6.2f} ms", flush=True)
