"""Stamped diagnostic build of the fused tail k_tail3 (full frames, folded compress_out): per-wave shader-clock sums per region of
the pipelined step, and the in-kernel clock.  usage: tail_stamps.py [h w] [totals]"""
import os, sys, ctypes
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
N = 8
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (N, 3, h, w)).astype(np.float32)).cuda()
lib = L.load()
lib.vsr_sr_tail_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
TOT = len(sys.argv) > 3 and sys.argv[3] == 'totals' or (len(sys.argv) == 2 and sys.argv[1] == 'totals')
nblk = N * ((w + 30) // 31)
buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
for _ in range(40): m(x)   # (warm the clock governor)
torch.cuda.synchronize()
L.check(lib.vsr_sr_tail_stamp_buffer(ctypes.c_void_p(buf.data_ptr()), 1 if TOT else 0))
m(x)
torch.cuda.synchronize()
lib.vsr_sr_tail_stamp_buffer(None, 0)
s = buf.view(nblk, 8, 8).double().cpu()
names = ["requests+tile3", "barrier", "A", "C", "B"]
print("shader cycles per steady step (median over workgroups), by wave:")
for wv in range(4):
    med = s[:, wv, :].median(dim=0).values
    n = max(med[5].item(), 1.0)
    print(f"  wave {wv}: " + "  ".join(f"{nm} {med[k].item()/n:7.1f}" for k, nm in enumerate(names)) + f"   steps {n:.0f}  loop/rows {med[6].item()/(h+3):7.1f}")
tot = s[:, :4, 6].median().item(); rt = s[:, :4, 7].median().item()
print(f"in-kernel clock: {tot / rt * 100:.0f} MHz  (loop {tot:.0f} cycles, {rt/100:.1f} us)")
