"""Three recurrent VSR.forward calls at the headline size (after two untimed ones): the workload of tools/frame_pmc.sh."""
import os, sys
os.environ.setdefault('MIOPEN_FIND_MODE', '2'); os.environ.setdefault('MIOPEN_LOG_LEVEL', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = fill_module_(VSR(upscale_factor=scale).eval(), 0).cuda()
clip = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (7, h, w, 3)).astype(np.float32)).cuda()
est = None
for t in range(5):
    est, _ = m(clip[t:t + 3], None, None, est, train=False)
torch.cuda.synchronize()
print("done", float(est.mean()))
