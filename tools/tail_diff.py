"""Where k_tail3 (tail_build 3) and k_tail (tail_build 1) differ: max |diff| per LR column strip and per HR row (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 960)
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (8, 3, h, w)).astype(np.float32)).cuda()
for fold in (True, False):
    m.fold_tail = fold
    m.tail_build = 1
    ref = m(x).clone()
    m.tail_build = 3
    got = m(x)
    d = (got - ref).abs()[0].amax(0).cpu().numpy()          # [4h, 4w]
    print(f"fold={fold}: max diff {d.max():.4g}")
    cols = d.max(0).reshape(-1)
    bad = np.nonzero(cols > 1e-3)[0]
    print("  bad HR columns:", bad[:40], "... count", len(bad), " strips:", sorted(set((bad // 4) // 31))[:40])
    rows = np.nonzero(d.max(1) > 1e-3)[0]
    print("  bad HR rows:", rows[:64], "count", len(rows))
