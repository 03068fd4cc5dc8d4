"""Per-layer time of the three guidance trunks at the bench size (HIP events around every convolution launch,
one stream): where the trunk milliseconds go.  usage: probe_layers.py [flow|depth|vos] [top] [h w]"""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "flow"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
h, w = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (540, 960)
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
fn = {"flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx),
      "depth": lambda: hx(fr), "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}[which]
for _t in os.environ.get('VSR_TUNING', '').split(','):
    if _t: L.load().vsr_conv2d_tuning(int(_t))
for _ in range(2): fn()
torch.cuda.synchronize()
L.TIMER.enabled = True; L.TIMER.reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
L.TIMER.enabled = False
S = L.TIMER.summary()
tot = sum(n * ms for n, ms in S.values())
print(f"{which}: {e0.elapsed_time(e1):.2f} ms wall with events, {tot:.2f} ms inside {sum(n for n, _ in S.values())} conv launches")
def flops(name, n):
    import re
    mm = re.match(r"conv N(\d+) (\d+)x(\d+) c(\d+)->(\d+) k(\d+)x(\d+) s(\d+)", name)
    if mm:
        N, H, W, ci, co, kh, kw, st = map(int, mm.groups())
        return 2.0 * N * (H // st) * (W // st) * ci * co * kh * kw
    mm = re.match(r"deconv4s2 N(\d+) (\d+)x(\d+) c(\d+)->(\d+)", name)
    N, H, W, ci, co = map(int, mm.groups())
    return 2.0 * N * H * W * ci * co * 16
for name, (n, ms) in sorted(S.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:top]:
    print(f"{name:46s} x{n:3d}  {ms*1e3:8.1f} us each  {n*ms:7.3f} ms  {flops(name, n) / (ms * 1e-3) / 1e12:7.1f} TFLOP/s")
