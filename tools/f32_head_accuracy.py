"""Error of the three float32 implementations of predict_flow-shaped layers against float64 (max |diff| / max |ref|)."""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from video_super_resolution_amd import trunk_f32
torch.set_grad_enabled(False)
torch.manual_seed(0)
for (N, C, H, W, Co) in [(2, 1024, 8, 15, 2), (2, 1026, 16, 30, 2), (2, 770, 32, 60, 2), (2, 386, 64, 120, 2), (2, 194, 128, 240, 2), (2, 128, 128, 240, 2), (1, 64, 64, 64, 1)]:
    x = torch.randn(N, C, H, W, device="cuda").relu_()      # (post-LeakyReLU-like activations: mostly positive, so the sum does not cancel)
    w = torch.randn(Co, C, 3, 3, device="cuda") / (C * 9) ** 0.5
    b = torch.randn(Co, device="cuda")
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    rng = ref.abs().max().item()
    wp = trunk_f32._pack(w)
    res = {}
    res["stock"] = F.conv2d(x, w, b, padding=1)
    res["flat/head (route 1)"] = trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, 3, 3, 1, 1, 1, 1)
    res["spatial (route 2)"] = trunk_f32.conv2d_fused(x, wp, None, b, False, 0.0, Co, 3, 3, 1, 1, 1, 2)
    cpu = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1).cuda()
    res["cpu float32"] = cpu
    print(f"N{N} c{C}->{Co} {H}x{W}: " + "   ".join(f"{k} {((v.double() - ref).abs().max().item() / rng):.2e}" for k, v in res.items()) +
          f"   stock vs cpu {((res['stock'] - cpu).abs().max().item() / rng):.2e}", flush=True)
