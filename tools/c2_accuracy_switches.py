"""Accuracy of the float32 configuration against the oracle (bench.py's 64 x 64 tile) under each of the float32 path's switches:
isolates which component moves the frame when the discrete guidance planes flip.  usage: c2_accuracy_switches.py"""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from video_super_resolution_amd import VSR, trunk_f32
from video_super_resolution_amd.weights import fill_module_
dev = torch.device("cuda")
model = fill_module_(VSR(upscale_factor=2).eval(), seed=0).to(dev)
model.precision = model.model.precision = "fp32"
first = None
def run(tag):
    global first
    if first is None:
        from oracle import vsr_oracle
        torch.set_num_threads(16); torch.set_flush_denormal(True)
        P = {k: v.detach() for k, v in fill_module_(VSR(upscale_factor=2).eval(), seed=0).state_dict().items()}
        data = torch.from_numpy(bench.synthetic_clip(0, 3, 64, 64))
        with torch.no_grad():
            first = (64, data, vsr_oracle.vsr_forward(P, data, None, upscale_factor=2))
    a = bench.accuracy_vs_oracle(model, 2, "fp32", dev, first=first)
    print(f"{tag:34s} psnr {a['psnr_vs_oracle_db']:7.2f} dB  max {a['max_rel_err']:.3e}  p99 {a['p99_rel_err']:.3e}  median {a['median_rel_err']:.3e}", flush=True)
run("defaults")
trunk_f32.FUSE = False; run("FUSE off"); trunk_f32.FUSE = True
trunk_f32.SPATIAL = False; run("SPATIAL off"); trunk_f32.SPATIAL = True
model.model.fuse_dt_f32 = False; run("fuse_dt_f32 off"); model.model.fuse_dt_f32 = True
model.share_planes = False; run("share_planes off"); model.share_planes = True
trunk_f32.ENABLED = False; run("trunks on stock operators"); trunk_f32.ENABLED = True
trunk_f32.ENABLED = False; model.model.fuse_dt_f32 = False; model.share_planes = False; run("all three off")
model.model.fuse_dt_f32 = False; model.share_planes = False; trunk_f32.ENABLED = True
run("trunks own, SR switches off")
orig = trunk_f32._route
trunk_f32._route = lambda N, C, H, W, Co, kh, kw, s, py, px: (0 if Co <= 4 else orig(N, C, H, W, Co, kh, kw, s, py, px)); run(".. heads on the stock operator")
trunk_f32._route = lambda N, C, H, W, Co, kh, kw, s, py, px: (0 if orig(N, C, H, W, Co, kh, kw, s, py, px) == 2 else orig(N, C, H, W, Co, kh, kw, s, py, px)); run(".. spatial layers on stock")
trunk_f32._route = lambda N, C, H, W, Co, kh, kw, s, py, px: (0 if orig(N, C, H, W, Co, kh, kw, s, py, px) == 1 else orig(N, C, H, W, Co, kh, kw, s, py, px)); run(".. flat layers on stock")
trunk_f32._route = lambda N, C, H, W, Co, kh, kw, s, py, px: (0 if (orig(N, C, H, W, Co, kh, kw, s, py, px) == 1 and kh == 1) else orig(N, C, H, W, Co, kh, kw, s, py, px)); run(".. flat 1x1 layers on stock")
trunk_f32._route = orig
