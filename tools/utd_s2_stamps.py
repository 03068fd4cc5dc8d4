"""Phase stamps of the software-pipelined x2 stage kernel k_utd_s2p (variant 3): cycles per steady iteration in the deconvolution slots,
the convolution slots, the stores behind them and at the barrier.  usage: utd_s2_stamps.py [h w planes]"""
import os, sys, ctypes
os.environ.setdefault("VSR_USE_XCHECK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920, 5)
m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), 0, "model.").cuda()
st = m._packed()["stage"][0]
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
lib = L.load()
buf = torch.zeros(64 * 64 * N * 4 * 8, dtype=torch.int64, device="cuda")
L.check(lib.vsr_sr_utd_s2_stamp_buffer(ctypes.c_void_p(buf.data_ptr())))
lib.vsr_sr_utd_s2_variant(3)
for _ in range(20): st(a, m._chain)
torch.cuda.synchronize(); buf.zero_(); st(a, m._chain); torch.cuda.synchronize()
lib.vsr_sr_utd_s2_variant(0); lib.vsr_sr_utd_s2_stamp_buffer(None)
s = buf.view(-1, 8).double().cpu()
s = s[s[:, 4] > 0]
per = s[:, :4] / s[:, 4:5]
med = per.median(0).values
print(f"{s.shape[0]} waves, {s[:, 4].median():.0f} steady iterations each; s_memtime ticks per iteration, median over waves:")
print(f"deconvolution slots {med[0]:.0f}   convolution slots {med[1]:.0f}   stores {med[2]:.0f}   barrier {med[3]:.0f}   sum {med.sum():.0f}")
