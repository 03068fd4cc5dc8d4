"""Trunk wall times under the kernel-selection modes of vsr_conv2d_tuning, one process, interleaved rounds:
usage trunks_tuning_ab.py [h w] -- 0 heuristic, 2 patch kernel whenever legal, 5 heuristic without k_conv_patch_r8, 10 / 11 128-channel
gather tiles always / never."""
import os, sys
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
m = fill_module_(VSR().eval(), 0).cuda()
fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (4, h, w, 3)).astype(np.float32)).cuda()
fx, hx, ox = m._flow_exec.get(), m._depth_exec.get(), m._vos_exec.get()
stages = {"flow (2 pairs)": lambda: m.FlowModule.forward_pairs([(fr[0], fr[1]), (fr[1], fr[2])], fx), "depth x4": lambda: hx(fr),
          "depth x1": lambda: hx(fr[:1]), "vos": lambda: m.VOSModule(fr[0], fr[1], ox)}
modes = [int(a) for a in os.environ.get("VSR_MODES", "0,2,5,10,11").split(",")]
res = {(s, md): [] for s in stages for md in modes}
lib = L.load()
for rnd in range(3):
    for md in modes:
        lib.vsr_conv2d_tuning(md)
        for name, fn in stages.items():
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res[(name, md)].append(e0.elapsed_time(e1) / 5)
lib.vsr_conv2d_tuning(0)
print(f"{h}x{w}: median of 3 rounds, ms   " + "  ".join(f"mode {md:2d}" for md in modes))
for name in stages:
    print(f"{name:16s} " + "  ".join(f"{sorted(res[(name, md)])[1]:7.3f}" for md in modes))
