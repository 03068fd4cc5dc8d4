"""Per-layer sweep of the convolution launch configurations at the benchmark size: every distinct layer of the three guidance
trunks (labels from the event timer) is run STANDALONE (random operands of the layer's shape) through the gather kernel and
through the tile kernel (conv_tile.hip) with forced tile widths / split counts, rounds interleaved in one process.
usage: tile_sweep.py [flow|depth|vos|all] [min_us]   -> table: layer, current route + us, best candidate + us"""
import os, re, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L, igemm
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
lib = L.load()
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
fns = {"flow": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth": lambda: hx(fr), "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}
labels = {}
for name, fn in fns.items():
    if which not in ("all", name): continue
    fn(); torch.cuda.synchronize()
    L.ROUTES.calls = []; L.ROUTES.enabled = True; fn(); L.ROUTES.enabled = False
    for lab, route in L.ROUTES.calls:
        labels.setdefault(lab, [0, route, name]); labels[lab][0] += 1
def tune(seq):
    for t in seq: lib.vsr_conv2d_tuning(t)
DEFAULT = (0, 2001, 4000, 5000, 1128)
def make(lab):
    mm = re.match(r"conv N(\d+) (\d+)x(\d+) c(\d+)->(\d+) k(\d+)x(\d+) s(\d+)$", lab)
    rs = np.random.RandomState(1)
    if mm:
        N, H, W, ci, co, kh, kw, st = map(int, mm.groups())
        if kh != kw: return None
        wt = torch.from_numpy((rs.randn(co, ci, kh, kw) / np.sqrt(ci * kh * kw)).astype(np.float32)).cuda()
        conv = igemm.HConv(wt, torch.zeros(co, device="cuda"), stride=st, pad=kh // 2, act=igemm.ACT_LEAKY)
        x = torch.from_numpy(rs.randn(N, H, W, ci).astype(np.float32)).cuda().half()
        fl = 2.0 * N * (H // st) * (W // st) * ci * co * kh * kw
        return (lambda: conv(x)), fl
    mm = re.match(r"deconv4s2 N(\d+) (\d+)x(\d+) c(\d+)->(\d+)$", lab)
    if mm:
        N, H, W, ci, co = map(int, mm.groups())
        wt = torch.from_numpy((rs.randn(ci, co, 4, 4) / np.sqrt(ci * 4)).astype(np.float32)).cuda()
        dc = igemm.HDeconv4s2(wt, torch.zeros(co, device="cuda"), act=igemm.ACT_LEAKY)
        x = torch.from_numpy(rs.randn(N, H, W, ci).astype(np.float32)).cuda().half()
        return (lambda: dc(x)), 2.0 * N * H * W * ci * co * 16
    return None
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
cands = {"cur": DEFAULT, "gather": (0, 2000, 1128), "gather_ns": (0, 2000, 1000)}
for bn in (64, 128):
    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
        cands[f"tile{bn}x{sp}"] = (0, 2003, 4000 + bn, 5000 + sp)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
tot_cur = tot_best = 0.0
print(f"{'layer':44s} {'n':>3s} {'route now':26s} {'us now':>8s}  {'best':>12s} {'us':>8s} {'TF/s':>7s}   gather  tile64x1 tile128x1")
for lab, (n, route, trunk) in sorted(labels.items(), key=lambda kv: kv[0]):
    made = make(lab)
    if made is None: continue
    fn, fl = made
    res = {}
    for rnd in range(3):
        for cn, seq in cands.items():
            tune(seq)
            fn(); 
            t = timeit(fn)
            if cn == "cur" and rnd == 0: r_now = lib.vsr_last_route().decode()
            res[cn] = min(res.get(cn, 1e9), t)
    tune(DEFAULT)
    if res["cur"] < min_us: continue
    best = min(res, key=res.get)
    tot_cur += n * res["cur"]; tot_best += n * res[best]
    print(f"{lab:44s} {n:3d} {r_now:26s} {res['cur']:8.1f}  {best:>12s} {res[best]:8.1f} {fl / res[best] / 1e6:7.0f}   {res['gather']:6.1f}  {res['tile64x1']:7.1f} {res['tile128x1']:8.1f}")
print(f"total per trunk call: now {tot_cur / 1e3:.3f} ms, best-of-candidates {tot_best / 1e3:.3f} ms")
