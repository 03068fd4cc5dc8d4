"""Runs only the fused x2 stage kernel k_utd_s2 (8 planes of LR h x w, default the C3-B size 1080 x 1920) -- target for
rocprofv3 --pmc (tools/utd_pmc.sh s2) and a quick timing."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 1920
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if len(sys.argv) > 4: N = int(sys.argv[4])   # planes per launch (5: what both SR passes launch)
m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), 0, "model.").cuda()
st = m._packed()["stage"][0]
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
from video_super_resolution_amd import _lib as L
lib = L.load()
VARS = tuple(int(v) for v in os.environ.get("S2_VARIANTS", "0,1").split(","))   # (0: shipping build, 1: branch-free build)
POST = os.environ.get("S2_POST", "0") == "1"   # the launch with the next group's uptran 1x1 fused in (k_utd_s2<.., POST>; variant 0 only)
if POST:
    VARS = (0,)
    run = lambda: st(a, m._chain, post=True)[0]
else:
    run = lambda: st(a, m._chain)
outs = {}
res = {v: [] for v in VARS}
for v in VARS:
    lib.vsr_sr_utd_s2_variant(v)
    for _ in range(2): outs[v] = run().clone()
torch.cuda.synchronize()
for r in range(4):   # interleaved rounds on one device
    for v in VARS:
        lib.vsr_sr_utd_s2_variant(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / reps)
lib.vsr_sr_utd_s2_variant(0)
for v, name in ((0, "branches"), (1, "flat")):
    if v not in VARS: continue
    ms = sorted(res[v])[len(res[v]) // 2]
    print(f"k_utd_s2 [{name:8s}] {N}x{h}x{w}: {ms:.3f} ms  -> {N*h*w*155648/ms/1e9:.1f} TFLOP/s")
if len(VARS) > 1: print("bit-identical:", all(torch.equal(outs[VARS[0]], outs[v]) for v in VARS))
