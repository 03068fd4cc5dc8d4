"""Stage-by-stage wall times of one VSR.forward at a given LR size (prints progressively)."""
import sys, time, os
os.environ.setdefault('MIOPEN_FIND_MODE','2'); os.environ.setdefault('MIOPEN_LOG_LEVEL','2'); os.environ.setdefault('MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK','0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib
from video_super_resolution_amd.weights import fill_module_
h, w = int(sys.argv[1]), int(sys.argv[2])
def log(*a):
    print(f"[{time.time()-T0:7.2f}s]", *a, flush=True)
T0 = time.time()
torch.set_grad_enabled(False)
m = fill_module_(VSR().eval(), 0).cuda(); m.precision = m.model.precision = sys.argv[3] if len(sys.argv) > 3 else 'fp16'; log('model ready', m.model.precision)
d = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (3, h, w, 3)).astype(np.float32)).cuda()
def timed(name, fn, n=2):
    for i in range(n):
        torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize(); log(f"{name} run{i}: {1e3*(time.time()-t):.1f} ms")
    return r
timed("flow(f0,f1)", lambda: m.FlowModule(d[0], d[1]))
timed("depth.predict", lambda: m.DepthModule.predict(d[0]))
timed("vos", lambda: m.VOSModule(d[0], d[1]))
x8 = torch.cat([d.permute(0, 3, 1, 2)] * 2 + [d.permute(0, 3, 1, 2)[:2]], 0).contiguous()
_lib.TIMER.enabled = True
timed("sr(x8)", lambda: m.model(x8), n=2)
print(_lib.TIMER.summary(), flush=True)
_lib.TIMER.enabled = False
timed("vsr.forward", lambda: m(d, None, None, None, train=False), n=3)
log("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
