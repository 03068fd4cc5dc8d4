"""What this chip sustains on the fused SR stage's instruction shape: on ONE device, back to back, >= 2 s each on
random data: k_utd3 (5-plane launches as the forward issues them), a bare v_mfma_f32_16x16x32_f16 loop, the same loop with k_utd3's
304 VALU : 144 MFMA : 17 LDS mix and no global memory (tools/microbench/power_roofline.hip), k_utd3 again.  Wall TFLOP/s and the
in-kernel clock (s_memtime / s_memrealtime) of each.  usage: power_roofline.py [seconds per arm] -> text on stdout, JSON to argv[2]"""
import ctypes, json, os, sys, time
os.environ.setdefault("VSR_USE_XCHECK", "1")   # the switches / superseded builds used here live in libvsr_hip_xcheck.so
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from video_super_resolution_amd import SRProjectionModule, _lib as L
from video_super_resolution_amd.weights import fill_module_
torch.set_grad_enabled(False)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda", 0)
pr = ctypes.CDLL(os.path.join(ROOT, "tools", "microbench", "libpower_roofline.so"))
pr.pr_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rs = np.random.RandomState(0)
seed = np.empty(10240, dtype=np.uint32)
seed[:8192] = rs.randn(16384).astype(np.float16).view(np.uint32)        # MFMA fragments / LDS image: fp16 N(0,1)
seed[8192:] = rs.randint(0, 2 ** 32, 2048, dtype=np.uint64).astype(np.uint32)   # VALU inputs: random bits
seed_d = torch.from_numpy(seed.view(np.int32)).to(dev)
stamps = torch.zeros(512, dtype=torch.int64, device=dev)
sink = torch.zeros(256, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
TRIPS = 8000
FLOP_LAUNCH = 256 * 4 * TRIPS * 144 * 16384.0    # 256 CUs x 4 waves x trips x 144 MFMAs x (16 x 16 x 32 x 2)

def run_loop(mode):
    def fn():
        rc = pr.pr_run(mode, seed_d.data_ptr(), TRIPS, stamps.data_ptr(), sink.data_ptr(), st)
        assert rc == 0, rc
    return fn

N, h, w = 5, 540, 960
m = fill_module_(SRProjectionModule().eval(), 0, "model.").cuda()
P = m._packed()
a = (torch.randn(N, h, w, 32, device=dev) * 20).half()
lib = L.load()
UTD_FLOP = N * h * w * 294912.0

def utd():
    m._utd(a, P["utd"][0], N, h, w)

def timed(fn, secs):
    """back-to-back launches for `secs` seconds -> (launches, seconds) by HIP events"""
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); n = 0
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20): fn()
        n += 20
        torch.cuda.synchronize() if n % 400 == 0 else None   # (keep the queue bounded)
    e1.record(); torch.cuda.synchronize()
    return n, e0.elapsed_time(e1) * 1e-3

def loop_clock():
    s = stamps.view(256, 2).double().cpu()
    return float((s[:, 0] / s[:, 1]).median() * 100.0), float(s[:, 0].median() / TRIPS)

def utd_clock():
    nblk = N * 31 + 512
    buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
    L.check(lib.vsr_sr_utd_stamp_buffer(ctypes.c_void_p(buf.data_ptr())))
    L.check(lib.vsr_sr_utd_variant(4))          # the production loop with one stamp pair around it
    for _ in range(300): utd()                   # warm the governor on the stamped build
    torch.cuda.synchronize(); buf.zero_(); utd(); torch.cuda.synchronize()
    lib.vsr_sr_utd_variant(0); lib.vsr_sr_utd_stamp_buffer(None)
    s = buf.view(nblk, 8, 8).double().cpu()
    s = s[s[:, 0, 7] > 0]
    return float((s[:, :4, 6] / s[:, :4, 7]).median() * 100.0)

rows = []
def arm(name, fn, flop, clock_fn):
    n, sec = timed(fn, SECS)
    clk = clock_fn()
    tf = flop * n / sec / 1e12
    rows.append(dict(arm=name, launches=n, seconds=round(sec, 3), tflops=round(tf, 1), in_kernel_clock_mhz=round(clk[0] if isinstance(clk, tuple) else clk, 0),
                     cycles_per_trip=round(clk[1], 1) if isinstance(clk, tuple) else None))
    print(f"{name:58s} {n:6d} launches {sec:6.2f} s  {tf:8.1f} TFLOP/s  in-kernel clock {rows[-1]['in_kernel_clock_mhz']:.0f} MHz"
          + (f"  {clk[1]:.1f} cycles per 144-MFMA trip" if isinstance(clk, tuple) else ""), flush=True)

print(f"# {torch.cuda.get_device_name(0)}; {SECS} s per arm, back to back on one device; random operands everywhere")
arm("k_utd3, 5 planes x 540 x 960 (the forward's launch)", utd, UTD_FLOP, utd_clock)
arm("bare v_mfma_f32_16x16x32_f16, one wave per SIMD", run_loop(0), FLOP_LAUNCH, loop_clock)
arm("144 MFMA + 304 VALU + 17 LDS per trip (k_utd3's mix), no global memory", run_loop(1), FLOP_LAUNCH, loop_clock)
arm("the same with 250 VALU per trip (a leaner loop)", run_loop(2), FLOP_LAUNCH, loop_clock)
arm("the same with 200 VALU per trip", run_loop(3), FLOP_LAUNCH, loop_clock)
arm("k_utd3 again", utd, UTD_FLOP, utd_clock)
bare, mix = rows[1]["tflops"], rows[2]["tflops"]
k3 = max(rows[0]["tflops"], rows[-1]["tflops"])
res = dict(device=torch.cuda.get_device_name(0), seconds_per_arm=SECS, arms=rows, bare_mfma_tflops=bare, mix_loop_tflops=mix, k_utd3_tflops=k3,
           mix_250_valu_tflops=rows[3]["tflops"], mix_200_valu_tflops=rows[4]["tflops"],
           k_utd3_of_mix=round(k3 / mix, 4), mix_of_spec_peak=round(mix / 2500.0, 4), bare_of_spec_peak=round(bare / 2500.0, 4),
           note="mix_loop_tflops = what this chip sustained on k_utd3's MFMA shape and VALU / LDS density with every operand in registers / "
                "LDS and no global memory: the practical peak bench.py reports beside the 2.5 PFLOP/s spec figure")
print(f"k_utd3 = {k3 / mix:.3f} of the mix loop, {k3 / bare:.3f} of the bare loop; the mix loop = {mix / 2500:.3f}, the bare loop = {bare / 2500:.3f} of the 2.5 PFLOP/s spec peak")
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
