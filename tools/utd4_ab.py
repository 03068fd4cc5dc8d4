"""Same-device A/B: the fused stage on k_utd3 (16x16x32 MFMA) and k_utd4 (32x32x16), plain and with the fused uptran slice, UTD_N planes
(default 5) x 540 x 960, interleaved rounds; then the whole SR forward under both builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
m.precision = "fp16"
P = m._packed()
x = torch.randint(0, 256, (8, 3, h, w), device="cuda").float()
variants = {}
for N in [int(v) for v in os.environ.get("UTD_N", "5,4,8").split(",")]:
    a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
    variants[f"k_utd3       {N} planes"] = (lambda a=a, N=N: m._utd(a, P["utd"][0], N, h, w), N)
    variants[f"k_utd4       {N} planes"] = (lambda a=a, N=N: m._utd4(a, P["utd4"][0], N, h, w), N)
    variants[f"k_utd3 +post {N} planes"] = (lambda a=a, N=N: m._utd_post(a, P["utd_post"][0], N, h, w), N)
    variants[f"k_utd4 +post {N} planes"] = (lambda a=a, N=N: m._utd4(a, P["utd4"][0], N, h, w, post=True), N)


def fwd(build):
    def f():
        m.utd_build = build
        return m(x)
    return f


variants["SR forward (8 planes), k_utd3"] = (fwd(3), 0)
variants["SR forward (8 planes), k_utd4"] = (fwd(4), 0)
for fn, _ in variants.values():
    for _ in range(2):
        fn()
torch.cuda.synchronize()
res = {k: [] for k in variants}
for r in range(5):
    for k, (fn, _) in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / reps)
for k, v in res.items():
    N = variants[k][1]
    ms = sorted(v)[len(v) // 2]
    print(f"{k:32s} {h}x{w}: median {ms:.4f} ms  (min {min(v):.4f})" + (f"  {N * h * w * 294912 / ms / 1e9:7.1f} TFLOP/s  {N * h * w * 294912 / ms / 1e9 / 2500:.3f} of peak" if N else ""))
