"""Which torch (non-library) kernels each stage of the frame launches: torch.profiler per stage, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
x8 = torch.cat([fr[:3].permute(0, 3, 1, 2)] * 2 + [fr[:2].permute(0, 3, 1, 2)], 0).contiguous()
stages = {"flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth": lambda: hx(fr),
          "vos": lambda: m.VOSModule(fr[0], fr[1], ox), "sr": lambda: m.model(x8)}
only = sys.argv[1:] or list(stages)
for name in only:
    fn = stages[name]
    for _ in range(2): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    rows = [(e.key, e.count, e.device_time_total / 1e3) for e in prof.key_averages() if e.device_time_total > 0 and e.key.startswith("aten::")]
    rows.sort(key=lambda r: -r[2])
    print(f"== {name}: aten ops with device time")
    for k, c, t in rows[:18]:
        print(f"   {k:40s} x{c:4d} {t:8.3f} ms")
