"""The x2 tail alone at C3-B's size (5 planes of 1080 x 1920): the compress_out chain launch + k_tail_s2 against k_tail_s2<FOLD>
(vsr_sr_tail_s2_fold_f16).  usage: tail_s2_fold_ab.py [planes h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
N, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 1080, 1920)
m = fill_module_(VSR(upscale_factor=2).eval(), 0).cuda().model
m.precision = "fp16"
P = m._packed()
rs = np.random.RandomState(0)
live = {k: torch.from_numpy((rs.randn(N, h * w, 32) * 20).astype(np.float16)).cuda() for k in (3, 6)}
cmap = torch.from_numpy(rs.randn(h * w, 32).astype(np.float32)).cuda()
raw = torch.empty((N, 3, 2 * h, 2 * w), dtype=torch.float32, device="cuda")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
co = lambda: m._chain([dict(ins=[(live[k], P["co_w"], 32 * (k - 1)) for k in (3, 6)], bias=P["co_b"], slope=P["co_a"], cmap=cmap)], N, h * w, keep=[True])[0]
with torch.no_grad():
    hid = co().view(N, h, w, 32)
    fold = (live[3].view(N, h, w, 32), live[6].view(N, h, w, 32), cmap)
    for rep in range(2):
        tc = t(co); tt = t(lambda: m._tail_raw(hid, P, False, raw)); tf = t(lambda: m._tail_raw(fold[0], P, False, raw, fold=fold))
        print(f"chain {tc:.3f} ms + tail {tt:.3f} ms = {tc + tt:.3f} ms;  tail with the folded 1x1 {tf:.3f} ms  ({N} planes of {h} x {w})")
