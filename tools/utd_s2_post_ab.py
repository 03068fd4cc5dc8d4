"""The x2 stage alone at C3-B's size (5 planes of 1080 x 1920): k_utd_s2, then the chain launch of the next group's uptran slice, against
k_utd_s2<POST> (vsr_sr_utd_s2_post_f16) which does both.  usage: utd_s2_post_ab.py [planes h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
N, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 1080, 1920)
m = fill_module_(VSR(upscale_factor=2).eval(), 0).cuda().model
m.precision = "fp16"
P = m._packed(); st = P["stage"][0]
a = torch.from_numpy((np.random.RandomState(0).randn(N, h, w, 32) * 20).astype(np.float16)).cuda()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def chain(o):
    return m._chain([dict(ins=[(o.view(N, h * w, 32), P["ut_w"][3], 32 * 4)], prev=None, bias=P["ut_b"][3], slope=P["ut_a"][3])], N, h * w, keep=[True])[0]
with torch.no_grad():
    o = st(a, m._chain)
    for rep in range(2):
        ts = t(lambda: st(a, m._chain)); tc = t(lambda: chain(o)); tp = t(lambda: st(a, m._chain, post=True))
        print(f"stage {ts:.3f} ms + chain {tc:.3f} ms = {ts + tc:.3f} ms;  stage with the fused 1x1 {tp:.3f} ms  ({N} planes of {h} x {w})")
