"""What this chip sustains on the float32 configuration's MFMA (v_mfma_f32_32x32x2_f32) next to the two float32 block kernels of the SR
net and the thin spatial trunk kernel, on ONE device, back to back, >= 2 s each: a bare loop of the instruction (operands in registers,
one wave per SIMD: tools/microbench/power_roofline_f32.hip), vsr_sr_deconv_f32 (+ fused 1x1) and vsr_sr_conv_f32 at 8 x 540 x 960 x2,
the 64 -> 16 11x11 layer at 4 x 540 x 960.  usage: power_roofline_f32.py [seconds per arm [out.json]]
build first: hipcc -O3 -fPIC -shared --offload-arch=gfx950 -o tools/microbench/libpower_roofline_f32.so tools/microbench/power_roofline_f32.hip"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from video_super_resolution_amd import _lib as L, trunk_f32
from video_super_resolution_amd.sr import pack_dt_frags
torch.set_grad_enabled(False)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda", 0)
pr = ctypes.CDLL(os.path.join(ROOT, "tools", "microbench", "libpower_roofline_f32.so"))
pr.pr_run_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rs = np.random.RandomState(0)
seed_d = torch.from_numpy(rs.randn(2048).astype(np.float32)).to(dev)
stamps = torch.zeros(512, dtype=torch.int64, device=dev)
sink = torch.zeros(256, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
TRIPS = 2000
FLOP_LOOP = 256 * 4 * TRIPS * 144 * 4096.0    # 256 CUs x 4 waves x trips x 144 MFMAs x (32 x 32 x 2 x 2)

def loop():
    assert pr.pr_run_f32(seed_d.data_ptr(), TRIPS, stamps.data_ptr(), sink.data_ptr(), st) == 0

lib = L.load()
N, h, w, S, K = 8, 540, 960, 2, 6
x = torch.from_numpy(rs.randn(N, 32, h, w).astype(np.float32)).to(dev)
wp = torch.from_numpy((rs.randn(K, K, 32, 32) / (4.0 * K)).astype(np.float32)).to(dev)
b = torch.from_numpy(rs.randn(32).astype(np.float32)).to(dev)
wdt = torch.from_numpy((rs.randn(32, 32) / 6.0).astype(np.float32)).to(dev)
fr = pack_dt_frags(wdt, 0)
hr = torch.empty((N, 32, S * h, S * w), dtype=torch.float32, device=dev)
lr = torch.empty((N, 32, h, w), dtype=torch.float32, device=dev)
BLOCK_FLOP = 2.0 * N * h * w * 32 * 32 * K * K
DT_FLOP = 2.0 * N * S * S * h * w * 32 * 32

def deconv():
    L.check(lib.vsr_sr_deconv_f32(L.dptr(x), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(hr), N, h, w, S, L.dptr(fr), L.dptr(b), L.cf(0.3), L.stream()))

def conv():
    L.check(lib.vsr_sr_conv_f32(L.dptr(hr), L.dptr(wp), L.dptr(b), L.cf(0.2), L.dptr(lr), N, h, w, S, L.stream()))

xt = torch.randn(4, 64, h, w, device=dev)
wt = torch.randn(16, 64, 11, 11, device=dev) / 88.0
wtp = trunk_f32._pack(wt)
ot = torch.empty((4, 16, h, w), device=dev)
THIN_FLOP = 2.0 * 4 * h * w * 64 * 16 * 121

def thin():
    trunk_f32.conv2d_fused(xt, wtp, None, None, False, 0.0, 16, 11, 11, 1, 5, 5, 2, out=ot)

def timed(fn, secs):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); n = 0
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(10): fn()
        n += 10
        torch.cuda.synchronize() if n % 200 == 0 else None
    e1.record(); torch.cuda.synchronize()
    return n, e0.elapsed_time(e1) * 1e-3

rows = []
def arm(name, fn, flop, clock=False):
    n, sec = timed(fn, SECS)
    tf = flop * n / sec / 1e12
    row = dict(arm=name, launches=n, seconds=round(sec, 3), tflops=round(tf, 1))
    if clock:
        s = stamps.view(256, 2).double().cpu()
        row["in_kernel_clock_mhz"] = round(float((s[:, 0] / s[:, 1]).median() * 100.0))
        row["cycles_per_mfma"] = round(float(s[:, 0].median() / TRIPS / 144), 1)
    rows.append(row)
    print(f"{name:64s} {n:6d} launches {sec:6.2f} s  {tf:7.1f} TFLOP/s" + (f"  in-kernel clock {row['in_kernel_clock_mhz']} MHz, {row['cycles_per_mfma']} cycles per MFMA" if clock else ""), flush=True)

print(f"# {torch.cuda.get_device_name(0)}; {SECS} s per arm, back to back on one device; random operands")
arm("bare v_mfma_f32_32x32x2_f32, one wave per SIMD", loop, FLOP_LOOP, clock=True)
arm("k_deconv_mfma_sh + fused 1x1, 8 x 540 x 960 x2", deconv, BLOCK_FLOP + DT_FLOP)
arm("k_conv_mfma_sh, 8 x 540 x 960 x2", conv, BLOCK_FLOP)
arm("k_conv_f32_sp16, 4 x 540 x 960 64 -> 16 11x11", thin, THIN_FLOP)
arm("bare loop again", loop, FLOP_LOOP, clock=True)
bare = max(rows[0]["tflops"], rows[-1]["tflops"])
res = dict(device=torch.cuda.get_device_name(0), seconds_per_arm=SECS, arms=rows, bare_mfma_f32_tflops=bare, bare_of_spec_peak=round(bare / 157.3, 4),
           deconv_of_bare=round(rows[1]["tflops"] / bare, 4), conv_of_bare=round(rows[2]["tflops"] / bare, 4), thin_of_bare=round(rows[3]["tflops"] / bare, 4),
           note="bare_mfma_f32_tflops = what this chip sustained on v_mfma_f32_32x32x2_f32 with every operand in registers: the practical peak "
                "bench.py reports for the float32 configuration beside the 157.3 TFLOP/s spec figure")
print(f"bare loop = {bare / 157.3:.3f} of the 157.3 TFLOP/s spec peak; deconv + 1x1 {res['deconv_of_bare']:.3f}, conv {res['conv_of_bare']:.3f}, thin spatial kernel {res['thin_of_bare']:.3f} of the bare loop")
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
