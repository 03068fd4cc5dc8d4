"""Same-device A/B of the x2 fused stage: k_utd_s2 (16x16x32 MFMA, two workgroups per CU) against k_utd_s2w (32x32x16, one wave per SIMD),
UTD_N planes (default 5) of LR 1080 x 1920 (the C3-B launch), interleaved rounds; prints the max difference between the two outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mods = {}
for build in (1, 2):
    m = fill_module_(SRProjectionModule(upscale_factor=2).eval(), 0, "model.").cuda()
    m.precision = "fp16"
    m.utd_s2_build = build
    mods[build] = (m, m._packed()["stage"][0])
res, outs = {}, {}
for N in [int(v) for v in os.environ.get("UTD_N", "5,3").split(",")]:
    a = (torch.randn(N, h, w, 32, device="cuda") * 20).half()
    for build, (m, st) in mods.items():
        for _ in range(2):
            outs[build] = st(a, m._chain).clone()
    torch.cuda.synchronize()
    d = (outs[1].float() - outs[2].float()).abs()
    print(f"N={N}: max |k_utd_s2 - k_utd_s2w| = {d.max().item():.4g} of range {outs[1].float().abs().max().item():.4g}; values that differ: {(d > 0).float().mean().item():.3f}")
    for r in range(4):
        for build, (m, st) in mods.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                st(a, m._chain)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((N, build), []).append(e0.elapsed_time(e1) / reps)
for (N, build), v in res.items():
    ms = sorted(v)[len(v) // 2]
    print(f"{'k_utd_s2 ' if build == 1 else 'k_utd_s2w'} {N}x{h}x{w}: {ms:.3f} ms  -> {N * h * w * 155648 / ms / 1e9:.1f} TFLOP/s  {N * h * w * 155648 / ms / 1e9 / 2500:.3f} of peak")
