"""The float32 deconvolution with the downtran 1x1 in its epilogue (k_deconv_mfma_sh<6, 2, 2, true>) at the C2 launch geometry (8 planes of
540 x 960, x2), a few launches: workload for tools/hbm_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import _lib as L
from video_super_resolution_amd.sr import pack_dt_frags
torch.set_grad_enabled(False)
N, h, w, S, K = 8, 540, 960, 2, 6
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(N, 32, h, w).astype(np.float32)).cuda()
wp = torch.from_numpy((rs.randn(K, K, 32, 32) / (4.0 * K)).astype(np.float32)).cuda()
b = torch.from_numpy(rs.randn(32).astype(np.float32)).cuda()
fr = pack_dt_frags(torch.from_numpy((rs.randn(32, 32) / 6.0).astype(np.float32)).cuda(), 0)
hr = torch.empty((N, 32, S * h, S * w), dtype=torch.float32, device="cuda")
lib = L.load()
for _ in range(6):
    L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(hr), N, h, w, S, L.dptr(fr), L.dptr(b), L.cf(0.3), L.stream()))
torch.cuda.synchronize()
print("ok")
