"""Partial-line writes: a thin convolution (16 / 1 out-channels) writing its 32 / 2 bytes per pixel into a channel slice of a
256-channel pixel row (512-byte stride) against the same layer writing a DENSE 16- / 1-channel tensor.  usage: thin_out_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import igemm
torch.set_grad_enabled(False)
rs = np.random.RandomState(0)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N, H, W = 4, 540, 960
buf = (torch.randn(N, H, W, 256, device="cuda") * 0.5).half()
for k, cout in ((3, 16), (7, 16), (11, 16), (3, 1)):
    w = torch.from_numpy((rs.randn(cout, 64, k, k) / np.sqrt(64 * k * k)).astype(np.float32)).cuda()
    b = torch.zeros(cout, device="cuda")
    conv = igemm.HConv(w, b, pad=k // 2, act=igemm.ACT_RELU)
    dense = torch.empty((N, H, W, cout), dtype=torch.float16, device="cuda")
    wide = torch.empty((N, H, W, 32), dtype=torch.float16, device="cuda")
    src = buf if cout == 16 else buf[..., :64].contiguous()
    us_slice = t(lambda: conv(buf, out=buf, out_coff=208 + 16 * (k % 3), in_coff=0)) if cout == 16 else t(lambda: conv(src, out=wide, out_coff=0))
    us_dense = t(lambda: conv(src if cout == 1 else buf, out=dense, out_coff=0, in_coff=0))
    print(f"{k}x{k} 64->{cout} at {N}x{H}x{W}: into the 512-byte pixel row {us_slice:7.1f} us, dense [N,H,W,{cout}] {us_dense:7.1f} us")
