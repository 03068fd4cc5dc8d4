"""Wall time of the four serial stages of VSR.forward (guidance 1, SR 1, guidance 2, SR 2), synchronising between them."""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from video_super_resolution_amd import VSR
from video_super_resolution_amd.vsr import maskprocess
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = 540, 960
m = fill_module_(VSR().eval(), 0).cuda()
d = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (3, h, w, 3)).astype(np.float32)).cuda()
est_img = None
for it in range(4):
    T = {}
    def tick(name, fn):
        torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize(); T[name] = 1e3 * (time.time() - t); return r
    f0, f1, f2 = d[0], d[1], d[2]
    frames = d.permute(0, 3, 1, 2)
    depth_cache = {}
    if est_img is None:
        est, est_hw3 = frames[0:1], f0
    else:
        est = F.interpolate(est_img.permute(0, 3, 1, 2), (h, w)); est_hw3 = est[0].permute(1, 2, 0).contiguous()
    pics, depth, _ = tick("guidance 1", lambda: m._guidance((f0, f1, f2), depth_cache, extra_depth=(est_hw3,)))
    mid = tick("SR 1", lambda: m.model(torch.cat((frames, pics, depth, est), 0), decimate=True))[0]
    def g2():
        mid_hw3 = mid.permute(1, 2, 0).contiguous()
        pics2, depth2, mask = m._guidance((est_hw3, mid_hw3, f2), depth_cache, with_vos=(est_hw3, mid_hw3))
        masked = torch.where(maskprocess(mask) != 0, torch.zeros_like(mid), mid).unsqueeze(0)
        return pics2, depth2, masked
    pics2, depth2, masked = tick("guidance 2", g2)
    out = tick("SR 2", lambda: m.model(torch.cat((frames, pics2, depth2, masked), 0)).permute(0, 2, 3, 1))
    est_img = out
    print("  ".join(f"{k} {v:6.2f} ms" for k, v in T.items()), f" sum {sum(T.values()):.2f}", flush=True)
