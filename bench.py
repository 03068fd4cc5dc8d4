#!/usr/bin/env python3
"""bench.py -- HR frames/s of the per-frame video-SR forward (`VSR.forward`, train=False) on MI355X.

Workload (BASELINE.json headline config, SURVEY.md 8(d) reading C3-A): synthetic clips of LR 540x960 frames,
x4 -> 2160x3840 HR frames, 3-frame window + recurrent estimate, seeded synthetic weights.  One "step" = one
VSR.forward call per rank (one output frame of that rank's clip); inputs are resident in HBM before the timed
region; `value` = frames produced by all ranks / max-over-ranks wall time.  Ranks own independent clips (weak
scaling, no data-path collective); the finished frames are gathered to rank 0 inside the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# MIOpen reads its find mode when the library is loaded (i.e. at `import torch`): set it first.  Without a gfx950
# find-db the default mode benchmarks every solver per new conv shape -- minutes of start-up at 540x960.
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
# the fast mode's AI ("TunaNet") solver predictor aborted the process twice in ~30 runs of the fp32 configuration
# (abort() inside torch conv -> MIOpen, no message); the plain heuristic fallback has not
os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32-input MFMA peak
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense


def synthetic_clip(clip_id: int, n_frames: int, h: int, w: int) -> np.ndarray:
    """Distribution 'S' of SURVEY.md 8(d): blurred noise scene translated by (2k, k) px per frame, 0..255."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(1234 + clip_id)
    pad = 4 * n_frames
    base = rs.uniform(0, 255, size=(h + pad, w + 2 * pad, 3)).astype(np.float32)
    base = gaussian_filter(base, sigma=(3, 3, 0))
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    frames = [np.floor(base[k:k + h, 2 * k:2 * k + w]) for k in range(n_frames)]
    return np.stack(frames).astype(np.float32)


def cpu_baseline(seconds_budget: float = 25.0):
    """The oracle (CPU restatement of the same forward) timed on this box's host cores on a bounded sample."""
    from oracle import vsr_oracle
    from video_super_resolution_amd import VSR
    from video_super_resolution_amd.weights import fill_module_
    # the GPU box gives one job a CPU share of 16 cores whatever os.cpu_count() says: never oversubscribe
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    torch.set_num_threads(cores)
    torch.set_flush_denormal(True)
    m = fill_module_(VSR().eval(), seed=0)
    P = {k: v.detach() for k, v in m.state_dict().items()}
    lr = 64
    data = torch.from_numpy(synthetic_clip(0, 3, lr, lr))
    t0 = time.time()
    with torch.no_grad():
        vsr_oracle.vsr_forward(P, data, None)
    dt = time.time() - t0
    return dict(seconds=dt, lr_px=lr * lr, cores=cores, sample=f"1 frame of VSR.forward at LR {lr}x{lr} (x4 -> {4*lr}x{4*lr})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--lr-h", type=int, default=540)
    ap.add_argument("--lr-w", type=int, default=960)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"],
                    help="SR stack: fp16 storage + fp32 accumulate on MFMA (headline config) or exact fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from video_super_resolution_amd import VSR, _lib
    from video_super_resolution_amd.distributed import gather_frames
    from video_super_resolution_amd.weights import fill_module_

    h, w = args.lr_h, args.lr_w
    model = fill_module_(VSR().eval(), seed=0).to(dev)
    model.precision = model.model.precision = args.precision
    n_frames = args.steps + args.warmup + 2
    clip = torch.from_numpy(synthetic_clip(rank, n_frames, h, w)).to(dev)  # resident in HBM before timing
    hf = torch.zeros((3, 4 * h, 4 * w, 3), dtype=torch.float32, device=dev)

    def progress(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    progress(f"model and clip resident on {torch.cuda.get_device_name(dev)}; warm-up")
    est = None
    finished = torch.empty((args.steps, 4 * h, 4 * w, 3), dtype=torch.float16, device=dev)  # output frames of the timed steps
    with torch.no_grad():
        # initialisation, not a step: both entry paths of forward (no estimate yet / recurrent estimate) run once so that
        # weight packing, executor construction and the caching allocator's first-touch hipMallocs are outside the clock
        prime = None
        for _ in range(2):
            prime, _ = model(clip[0:3], None, hf, prime, train=False)
        del prime
        torch.cuda.synchronize()
        for t in range(args.warmup):
            est, _ = model(clip[t:t + 3], None, hf, est, train=False)
            torch.cuda.synchronize()
            progress(f"warm-up frame {t} done")
        if world > 1:
            dist.barrier()
        # HIP events around the dominant kernel's launches only (the fused up->tran->down stage; the exact-fp32
        # configuration's counterpart is the k8 s4 conv / deconv pair): ~12 event pairs per frame, not one per launch
        _lib.TIMER.reset()
        _lib.TIMER.only = {"sr_utd_f16"} if args.precision == "fp16" else {"sr_conv8s4_f32", "sr_deconv8s4_f32"}
        _lib.TIMER.enabled = True
        t0 = time.perf_counter()
        for i, t in enumerate(range(args.warmup, args.warmup + args.steps)):
            est, _ = model(clip[t:t + 3], None, hf, est, train=False)
            finished[i].copy_(est[0])  # [K,4h,4w,3] fp16 (values are 0..255-ish pixels)
        gathered = gather_frames(finished, dst=0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        _lib.TIMER.enabled = False
        _lib.TIMER.only = None
        progress(f"timed region: {elapsed:.3f} s for {args.steps} steps")
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if rank == 0:
        assert gathered is not None and sum(g.shape[0] for g in gathered) == world * args.steps
        assert torch.isfinite(finished.float()).all()

    if rank == 0:
        fps = world * args.steps / elapsed
        # ---- roofline of the dominant kernel, timed with HIP events inside the timed region
        timers = _lib.TIMER.summary()
        dom = max(timers.items(), key=lambda kv: kv[1][0] * kv[1][1]) if timers else None
        roof = None
        if dom is not None:
            name, (launches, ms) = dom
            # algorithmic FLOPs per launch (per LR pixel per image, 8 images per launch):
            #   one k8 s4 (de)conv 32->32 = 131,072; the fused up->tran->down stage = 131,072 + 16*2,048 + 131,072
            per_px = 294912.0 if name == "sr_utd_f16" else 131072.0
            peak = FP16_MFMA_PEAK_TFLOPS if name.startswith("sr_utd_f16") else FP32_PEAK_TFLOPS
            flop = 8 * h * w * per_px
            achieved = flop / (ms * 1e-3) / 1e12
            # HBM bytes per launch of this kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
            # separate passes, gfx950 correction applied; profiles/r01_k_utd3_pmc.json) -- same launch geometry only
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_k_utd3_pmc.json")
            if name == "sr_utd_f16" and (h, w) == (540, 960) and os.path.exists(pmc):
                with open(pmc) as f:
                    traffic = json.load(f)["hbm"]["traffic_bytes_per_launch"]
            roof = dict(bound="mfma", kernel=name, achieved=round(achieved, 3), peak=peak, unit="TFLOP/s",
                        frac=round(achieved / peak, 4), traffic=traffic, launches_timed=launches, avg_ms=round(ms, 4),
                        algorithmic_flop_per_launch=flop)
        line = dict(metric="HR frames/sec, 1080p->4K x4 VSR (LR 540x960 -> 2160x3840), VSR.forward end-to-end",
                    value=round(fps, 4), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32" if args.precision == "fp32" else "f16", data="synthetic",
                    config=dict(workload=f"C3-A: LR {h}x{w} x4 -> {4*h}x{4*w}, 3-frame window + recurrent estimate, "
                                         f"one clip per GPU, seeded synthetic weights", precision=args.precision,
                                parallelism=f"clip-dp{world}"),
                    roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            progress("timing the CPU oracle on a 64x64 LR tile (about half a minute)")
            cb = cpu_baseline()
            px_per_s = cb["lr_px"] / cb["seconds"]
            line["cpu_baseline"] = dict(value=round(px_per_s / (h * w), 8), unit="frames/s", cores=cb["cores"], kind="port",
                                        sample=cb["sample"] + f", {cb['seconds']:.1f} s; scaled linearly in pixels to "
                                                              f"LR {h}x{w} (extrapolated)")
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
