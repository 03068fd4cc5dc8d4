"""Same-device A/B of the stage with the fused uptran 1x1 (vsr_sr_utd_post_f16) against the two launches it replaces
(vsr_sr_utd_f16 + the one-stage vsr_sr_chain1x1_f16), UTD_N planes (default 5) x 540 x 960, interleaved rounds; also the whole
SR forward (8 planes, 5 unshared) with VSR fuse_uptran on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
N, h, w = int(os.environ.get("UTD_N", "5")), int(sys.argv[1]) if len(sys.argv) > 1 else 540, int(sys.argv[2]) if len(sys.argv) > 2 else 960
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
m.precision = "fp16"
P = m._packed()
a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
hp = h * w


def two():
    o = m._utd(a, P["utd"][0], N, h, w)
    return m._chain([dict(ins=[(o.view(N, hp, 32), P["ut_w"][3], 128)], prev=None, bias=P["ut_b"][3], slope=P["ut_a"][3])], N, hp, keep=[True])[0]


variants = {"utd alone": lambda: m._utd(a, P["utd"][0], N, h, w), "utd + chain (two launches)": two,
            "utd_post (one launch)": lambda: m._utd_post(a, P["utd_post"][0], N, h, w)}
x = torch.randint(0, 256, (8, 3, h, w), device="cuda").float()


def fwd(flag):
    def f():
        m.fuse_uptran = flag
        return m(x)
    return f


variants["SR forward, uptran apart"] = fwd(False)
variants["SR forward, uptran fused"] = fwd(True)
for fn in variants.values():
    for _ in range(2):
        fn()
torch.cuda.synchronize()
res = {k: [] for k in variants}
for r in range(5):
    for k, fn in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / reps)
for k, v in res.items():
    print(f"{k:30s} {N}x{h}x{w}: median {sorted(v)[len(v) // 2]:.4f} ms  (min {min(v):.4f})")
