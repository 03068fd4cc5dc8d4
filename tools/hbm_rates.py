"""Calibration: what plain fill / copy / read kernels reach on this device (GB/s), to judge store-bound layers against."""
import torch
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (128, 512, 1024):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.float16, device="cuda"); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb:5d} MB: fill {mb/1024/tf*1e3:6.2f} TB/s   copy (read+write) {2*mb/1024/tc*1e3:6.2f} TB/s   read (sum) {mb/1024/tr*1e3:6.2f} TB/s")
