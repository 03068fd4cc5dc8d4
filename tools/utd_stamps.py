"""Stamped diagnostic builds of the fused stage: per-wave shader-clock sums per phase of the march, and the
in-kernel clock (s_memtime / s_memrealtime).  usage: utd_stamps.py [2|3|4]   (2: k_utd3, 3: k_utd, 4: k_utd3 totals only)"""
import os, sys, ctypes
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, 540, 960
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
lib = L.load()
nblk = N * 31
buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
L.check(lib.vsr_sr_utd_stamp_buffer(ctypes.c_void_p(buf.data_ptr())))
L.check(lib.vsr_sr_utd_variant(VAR))
for _ in range(200):   # warm the clock governor with back-to-back launches
    m._utd(a, P["utd"][0], N, h, w)
torch.cuda.synchronize()
buf.zero_()
m._utd(a, P["utd"][0], N, h, w)
torch.cuda.synchronize()
lib.vsr_sr_utd_variant(0)
lib.vsr_sr_utd_stamp_buffer(None)
s = buf.view(nblk, 8, 8).double().cpu()
if VAR == 3:
    names, nw, TOT, RT = ["P2 (down conv)", "P1 (deconv+1x1)", "reduce+LR store", "barrier"], 8, 4, 5
else:
    names, nw, TOT, RT = ["A", "B", "C", "D", "E", "partials+LR store+barrier"], 4, 6, 7
print("shader cycles per LR row (median over workgroups), by wave:")
for wv in range(nw):
    med = s[:, wv, :].median(dim=0).values
    print(f"  wave {wv}: " + "  ".join(f"{n} {med[k].item()/h:7.1f}" for k, n in enumerate(names)) + f"   loop {med[TOT].item()/h:7.1f}")
tot = s[:, :nw, TOT].median().item(); rt = s[:, :nw, RT].median().item()
print(f"in-kernel clock: {tot / rt * 100:.0f} MHz  (loop {tot:.0f} cycles, {rt/100:.1f} us)")
