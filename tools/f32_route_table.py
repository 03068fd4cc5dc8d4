"""Every convolution layer of one float32 VSR.forward (BASELINE config C2: LR 540 x 960, x2) with the route trunk_f32 gives it, and the
device time of each implementation on that layer's shape (spatial-reuse kernels / flat kernel / stock operator), ranked by the
routed time x the number of calls per frame.  usage: f32_route_table.py [lr_h lr_w]"""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import OrderedDict
import torch, torch.nn.functional as F
from video_super_resolution_amd import VSR, trunk_f32
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
dev = torch.device("cuda")
model = fill_module_(VSR(upscale_factor=2).eval(), seed=0).to(dev)
model.precision = model.model.precision = "fp32"
frames = torch.randint(0, 256, (3, h, w, 3), device=dev).float()
hf = torch.zeros((3, 2 * h, 2 * w, 3), device=dev)
est, _ = model(frames, None, hf, None, train=False)
seen = OrderedDict()
orig = trunk_f32._route
def rec(N, C, H, W, Co, kh, kw, stride, py, px):
    r = orig(N, C, H, W, Co, kh, kw, stride, py, px)
    key = (N, C, H, W, Co, kh, kw, stride, py, px)
    seen.setdefault(key, [r, 0])[1] += 1
    return r
trunk_f32._route = rec
model(frames, None, hf, est, train=False)
trunk_f32._route = orig
torch.cuda.synchronize()
for k in seen: seen[k][1] //= 1   # (a fused group asks once; a stock-routed layer asks twice: group, then the module itself)
def t(fn, reps=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = []
for (N, C, H, W, Co, kh, kw, s, py, px), (route, calls) in seen.items():
    if route == 0: calls = (calls + 1) // 2
    x = torch.randn(N, C, H, W, device=dev)
    wt = torch.randn(Co, C, kh, kw, device=dev) / (C * kh * kw) ** 0.5
    b = torch.randn(Co, device=dev)
    wp = trunk_f32._pack(wt)
    Ho, Wo = (H + 2 * py - kh) // s + 1, (W + 2 * px - kw) // s + 1
    out = torch.empty((N, Co, Ho, Wo), device=dev)
    flat = t(lambda: trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, kh, kw, s, py, px, 1, out=out))
    sp = None
    if s == 1 and kh * kw >= 9:
        try: sp = t(lambda: trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, kh, kw, s, py, px, 2, out=out))
        except RuntimeError: sp = None
    stock = t(lambda: F.conv2d(x, wt, b, stride=s, padding=(py, px)))
    routed = (stock, flat, sp if sp is not None else flat)[route]
    rows.append((routed * calls, N, C, H, W, Co, kh, kw, s, calls, route, sp, flat, stock))
    del x, wt, out
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows); best = sum(min(v for v in r[11:14] if v is not None) * r[9] for r in rows)
print(f"{len(rows)} distinct layers, {sum(r[9] for r in rows)} convolution calls per frame: routed {tot:.1f} ms, best-of-three {best:.1f} ms")
for (tt, N, C, H, W, Co, kh, kw, s, calls, route, sp, flat, stock) in rows:
    fl = 2.0 * N * ((H + 2 * (kh // 2) - kh) // s + 1) * ((W + 2 * (kw // 2) - kw) // s + 1) * C * Co * kh * kw
    f = lambda v: f"{v:7.3f}" if v is not None else "      -"
    bestv = min(v for v in (sp, flat, stock) if v is not None)
    mark = "" if (sp, flat, stock)[(2, 1, 0).index(route)] in (bestv, None) or (stock, flat, sp)[route] <= 1.05 * bestv else "   <-- not the fastest"
    print(f"N{N} {H}x{W} c{C}->{Co} k{kh}x{kw} s{s} x{calls}: routed {('stock', 'flat', 'spatial')[route]:7s} {tt:7.2f} ms ({fl * calls / tt / 1e9:5.1f} TFLOP/s)  "
          f"spatial {f(sp)}  flat {f(flat)}  stock {f(stock)}{mark}")
