"""Standalone wall time of each fp16 trunk executor at the bench size."""
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
def bench(name, fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {1e3*(time.time()-t)/n:.2f} ms", flush=True)
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
bench("FlowNet2 batch 2 (+flow2img)", lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx))
bench("hourglass batch 4", lambda: hx(fr))
bench("hourglass batch 1", lambda: hx(fr[:1]))
bench("OSVOS pair", lambda: m.VOSModule(fr[0], fr[1], ox))
x8 = torch.cat([fr[:3].permute(0, 3, 1, 2)] * 2 + [fr[:2].permute(0, 3, 1, 2)], 0).contiguous()
bench("SR call", lambda: m.model(x8))
