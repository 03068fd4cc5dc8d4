"""Speed of the stock-conv guidance trunks under dtype / memory-format choices (full LR size)."""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (1, 3, h, w)).astype(np.float32)).cuda()
def bench(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {1e3*(time.time()-t)/n:.2f} ms", flush=True)
import copy
for dt in (torch.float32, torch.float16, torch.bfloat16):
    for cl in (False, True):
        hg = copy.deepcopy(m.DepthModule.model.netG).to(dt)
        xi = x.to(dt)
        if cl:
            hg = hg.to(memory_format=torch.channels_last); xi = xi.contiguous(memory_format=torch.channels_last)
        bench(f"HG {dt} cl={cl}", lambda: hg(xi))
        vg = copy.deepcopy(m.VOSModule.net).to(dt)
        x2 = torch.cat([xi, xi], 0)
        if cl:
            vg = vg.to(memory_format=torch.channels_last); x2 = x2.contiguous(memory_format=torch.channels_last)
        bench(f"OSVOS {dt} cl={cl}", lambda: vg(x2))
        fs = copy.deepcopy(m.FlowModule.net.flownets_1).to(dt)
        x12 = torch.randn(1, 12, 512, 960, device='cuda', dtype=dt)
        if cl:
            fs = fs.to(memory_format=torch.channels_last); x12 = x12.contiguous(memory_format=torch.channels_last)
        bench(f"FlowNetS {dt} cl={cl}", lambda: fs(x12))
