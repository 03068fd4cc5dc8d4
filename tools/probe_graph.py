"""Whole-frame hipGraph capture of VSR.forward (recurrent variant) vs eager: wall time per frame."""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
d = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (3, h, w, 3)).astype(np.float32)).cuda()
est0, _ = m(d, None, None, None, train=False)
for _ in range(2):
    est0, _ = m(d, None, None, est0, train=False)
torch.cuda.synchronize()
t = time.time()
for _ in range(5):
    o, _ = m(d, None, None, est0, train=False)
torch.cuda.synchronize(); print(f"eager: {1e3*(time.time()-t)/5:.1f} ms/frame", flush=True)
sd, se = d.clone(), est0.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(sd, None, None, se, train=False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    so, _ = m(sd, None, None, se, train=False)
torch.cuda.synchronize(); print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replay vs eager max diff", (so - o).abs().max().item(), flush=True)
t = time.time()
for _ in range(10):
    g.replay()
torch.cuda.synchronize(); print(f"graph: {1e3*(time.time()-t)/10:.1f} ms/frame", flush=True)
