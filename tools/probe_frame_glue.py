"""aten ops with device time inside one VSR.forward frame (torch.profiler): the glue between the C-ABI launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
d = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (5, h, w, 3)).astype(np.float32)).cuda()
est = None
hf = [None, None, None]
for i in range(2):
    est, _ = m(d[i:i + 3], None, hf, est, train=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    est, _ = m(d[2:5], None, hf, est, train=False); torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total / 1e3) for e in prof.key_averages() if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
print(f"aten ops, self device time: total {sum(r[2] for r in rows):.3f} ms in {sum(r[1] for r in rows)} calls")
for k, c, t in rows[:25]:
    print(f"   {k:40s} x{c:4d} {t:8.3f} ms")
ev = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
for e in ev[:45]:
    st = [f for f in e.stack if "video_super_resolution_amd" in f]
    print(f"{e.key:28s} x{e.count:3d} {e.self_device_time_total/1e3:7.3f} ms  <- " + " <- ".join(x.split("video_super_resolution_amd/")[-1] for x in st[:3]))
