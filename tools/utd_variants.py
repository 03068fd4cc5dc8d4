"""Builds of the fused stage timed against each other, interleaved rounds in ONE process on ONE device (devices
differ by >10 % in sustained clock: never compare across gpurun boxes), plus a bit-identity check.
usage: utd_variants.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, 540, 960
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
lib = L.load()
names = {0: "k_utd3 (one wave per SIMD)", 1: "k_utd  (two waves per SIMD)"}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
outs = {}
for k in names:
    L.check(lib.vsr_sr_utd_variant(k)); outs[k] = m._utd(a, P["utd"][0], N, h, w).clone()
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(outs[0], outs[1])))
res = {k: [] for k in names}
for r in range(rounds):
    for k in names:
        lib.vsr_sr_utd_variant(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): m._utd(a, P["utd"][0], N, h, w)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 5)
lib.vsr_sr_utd_variant(0)
flop = 294912.0 * N * h * w
for k, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{names[k]:30s} median {med:.4f} ms  min {min(v):.4f}  {flop / med / 1e9:.0f} TFLOP/s")
