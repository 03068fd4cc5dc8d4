"""A/B of two builds of libvsr_hip.so on the fused stage kernel, interleaved rounds in ONE process on ONE device
(devices differ by >10 % in sustained clock: never compare across gpurun boxes).
usage: utd_ab.py libA.so libB.so [rounds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, 540, 960
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
out = torch.empty_like(a)
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
def run(lib, reps):
    for _ in range(reps):
        rc = lib.vsr_sr_utd_f16(L.dptr(a, torch.float16), L.dptr(P["utd"][0], torch.uint8), L.dptr(out, torch.float16), N, h, w, h, 0, 1, L.stream())
        assert rc == 0
for lib in libs: run(lib, 3)
torch.cuda.synchronize()
res = [[] for _ in libs]
for r in range(rounds):
    for i, lib in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib, 5); e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) / 5)
for p, r in zip(sys.argv[1:3], res):
    r = sorted(r)
    print(f"{os.path.basename(p):28s} median {r[len(r)//2]:.4f} ms  min {r[0]:.4f}  max {r[-1]:.4f}")
