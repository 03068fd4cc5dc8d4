#!/usr/bin/env python3
"""bench.py -- HR frames/s of the per-frame video-SR forward (`VSR.forward`, train=False) on MI355X.

Default workload = BASELINE.json's headline config, SURVEY.md 8(d) reading C3-A: synthetic clips of LR 540x960 frames,
x4 -> 2160x3840 HR frames, 3-frame window + recurrent estimate, seeded synthetic weights, fp16 storage / fp32 accumulate.
One "step" = one VSR.forward call per rank (one output frame of that rank's clip); inputs are resident in HBM before the
timed region; `value` = frames produced by all ranks / max-over-ranks wall time.  Ranks own independent clips (weak
scaling, no data-path collective); the finished frames are gathered to rank 0 inside the timed region.

    python bench.py                                   # C3-A on one GPU
    python bench.py --config C2|C3B|C5                # the x2 configurations of BASELINE.json (scale extension)
    python bench.py --precision fp32                  # C3-A in exact float32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--clips 32]      # C4: 32 clips round-robin, per-clip gathers

--config   LR -> HR                scale  precision  (BASELINE.json `configs`)
  C3A      540x960  -> 2160x3840    x4     fp16      headline, reference-native geometry (parity-pinned)
  C3B      1080x1920 -> 2160x3840   x2     fp16      label-faithful "1080p -> 4K" input size
  C2       540x960  -> 1080x1920    x2     fp32
  C5       2160x3840 -> 4320x7680   x2     fp16      depth + VOS guidance (always on), HBM stress
  (C1, the 128x128 CPU plumbing case, is `python -m video_super_resolution_amd.driver`; C4 = C3A with --clips 32 --gpus 8)
"""
import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import time

# MIOpen reads its find mode when the library is loaded (i.e. at `import torch`): set it first.  Without a gfx950
# find-db the default mode benchmarks every solver per new conv shape -- minutes of start-up at 540x960.
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
# the fast mode's AI ("TunaNet") solver predictor aborted the process twice in ~30 runs of the fp32 configuration
# (abort() inside torch conv -> MIOpen, no message); the plain heuristic fallback has not
os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
# HIP's default of four hardware queues is the measured optimum for this forward's four streams (3 queues -0.6 %; a fifth ACTIVE queue cost
# 20 % while the forward used five streams: LAB_NOTES R4.7); assigned unconditionally so that an inherited setting cannot move the
# number, and printed in the JSON line (`env`)
os.environ["GPU_MAX_HW_QUEUES"] = "4"
os.environ["HIP_FORCE_DEV_KERNARG"] = "1"   # (this image's default behaviour; = 0 costs the ~600 launches of a frame 2 %)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32-input MFMA peak
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense

CONFIGS = {   # name: (LR h, LR w, scale, precision, label)
    "C3A": (540, 960, 4, "fp16", "C3-A: LR 540x960 x4 -> 2160x3840"),
    "C3B": (1080, 1920, 2, "fp16", "C3-B: LR 1080x1920 x2 -> 2160x3840 (scale extension)"),
    "C2": (540, 960, 2, "fp32", "C2: LR 540x960 x2 -> 1080x1920 (scale extension)"),
    "C5": (2160, 3840, 2, "fp16", "C5: LR 2160x3840 x2 -> 4320x7680 with depth + VOS guidance (scale extension)"),
}

# FLOPs per LR pixel per plane ACTUALLY executed by the SR net (SURVEY.md App. C "F_min": zero-fill dataflow, constant
# branches cached, dead tails skipped): head 15,104 + 3 steps x (compress_in 4,096 + compress_out 12,288 +
# 2 x [uptran slice 2,048 + dc + downtran slice 2,048 s^2 + dc]) + last tail (dc + 1,728 s^2), dc = 2*32*32*k^2/... per LR px
SR_FLOP_PER_PX = {4: 2004736.0, 2: 1091072.0, 3: 1507264.0}
STAGE_FLOP_PER_PX = {4: 294912.0, 2: 155648.0, 3: 219136.0}     # one up -> tran -> down stage: dc + 2,048 s^2 + dc
TRUNK_FLOP_PER_PX = dict(flownet2=1.008e6, hourglass=1.227e6, osvos=0.637e6)   # SURVEY.md 6 (FlopCounterMode)


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


def executed_flop_per_frame(h, w, scale, shared_planes=3, precision="fp16"):
    """FLOPs one VSR.forward executes in this implementation: the SR net's head + FeedbackBlock on 8 planes in pass 1 and
    on the 8 - `shared_planes` planes pass 2 does not share with it (the three LR frames are evaluated once), its tail on
    8 planes in both passes (pass 1 at the pixels (s i, s j) only: ~0.6 of the tail's MFMAs) + 4 FlowNet2 runs on the
    64-aligned crop + 5 hourglass runs (f0, f1, f2, estimate, pass-1 frame: the reference's 8 runs have 4-5 distinct
    inputs) + 2 OSVOS runs."""
    crop = (h // 64) * 64 * ((w // 64) * 64)
    k = scale + 4
    tail = 2 * 32 * 32 * k * k + 1728 * scale ** 2            # `out` deconv + conv_out per LR pixel and plane
    trunk = SR_FLOP_PER_PX[scale] - tail                       # head + 3 FeedbackBlock steps
    sr = h * w * ((8 + 8 - shared_planes) * trunk + 8 * tail * (1.0 + 0.6))
    if precision == "fp32":   # the float32 path keeps the shared planes' PRE-FUSION maps: pass 2 skips their tail too; no decimated tail
        sr = h * w * (8 + 8 - shared_planes) * (trunk + tail)
    trunks = 4 * crop * TRUNK_FLOP_PER_PX["flownet2"] + 5 * h * w * TRUNK_FLOP_PER_PX["hourglass"] + \
        2 * h * w * TRUNK_FLOP_PER_PX["osvos"]
    return sr + trunks


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(scale: int, sizes=(64, 96, 128), budget_s: float = 75.0):
    """The oracle (CPU restatement of the same forward) timed on this box's host cores at LR 64^2, 96^2, 128^2 (one frame
    each); a least-squares line seconds = a + b * pixels gives the extrapolated headline-size figure.
    Larger tiles are skipped once the budget is spent (the fit then uses the points measured)."""
    from oracle import vsr_oracle
    from video_super_resolution_amd import VSR
    from video_super_resolution_amd.weights import fill_module_
    avail = host_cores()
    cores = max(1, min(16, avail))   # a gpurun job's CPU share is 16 cores whatever the affinity mask says
    torch.set_num_threads(cores)
    torch.set_flush_denormal(True)
    m = fill_module_(VSR(upscale_factor=scale).eval(), seed=0)
    P = {k: v.detach() for k, v in m.state_dict().items()}
    pts = []
    first = None
    t_start = time.time()
    for lr in sizes:
        if pts and (time.time() - t_start) + pts[-1][1] * (lr * lr) / pts[-1][0] > budget_s:   # next tile ~ linear in pixels
            break
        data = torch.from_numpy(synthetic_clip(0, 3, lr, lr))
        t0 = time.time()
        taps = {} if first is None else None   # (references to tensors the forward holds anyway: no extra work inside the timed call)
        with torch.no_grad():
            out = vsr_oracle.vsr_forward(P, data, None, upscale_factor=scale, taps=taps)
        pts.append((lr * lr, time.time() - t0))
        if first is None:
            first = (lr, data, out, taps)    # kept: the GPU forward on the same tile gives the line's accuracy fields
    px = np.array([p[0] for p in pts], dtype=np.float64)
    sec = np.array([p[1] for p in pts], dtype=np.float64)
    if len(pts) >= 2:
        b, a = np.polyfit(px, sec, 1)
    else:
        b, a = sec[0] / px[0], 0.0
    return dict(points=[(int(p), round(float(s), 2)) for p, s in pts], slope_s_per_px=float(b), intercept_s=float(a),
                cores=cores, cores_available=avail, cpu=cpu_model(), first=first)


# LR receptive radius of the SR net: conv_in 3x3 (1) + 3 steps x 2 up/down pairs (x4, k8 s4 p2: 1 LR pixel per pair; x2 k6 s2 p2 and x3 k7 s3 p2: 2)
# + out deconv, conv_out, bilinear skip (<= 3), rounded up
SR_LR_RADIUS = {4: 10, 3: 18, 2: 18}


def plane_flip_report(build_taps, ref_taps, got_hwc, ref_hwc, scale, radius=None):
    """The DISCRETE guidance planes of the two SR passes -- the uint8 flow pictures (planes 3, 4; flow_utils.py:4-24 via
    video_super_resolution.py:28-35,46-52) and the 0/1 VOS mask (:54,:58-60) -- compared pixel by pixel between two evaluations of one
    forward (`*_taps`: pass1_input / pass2_input [8,3,h,w], vos_mask), and the frame error OUTSIDE the receptive fields of the flipped
    pixels: pass-1 flips reach pass 2 through the decimated pass-1 frame, so the excluded set is dilate(flips2 | mask flips |
    dilate(flips1)), each dilation by the SR net's LR receptive radius.  got / ref: [H,W,3] frames (numpy float64).  -> dict"""
    import torch.nn.functional as F
    radius = SR_LR_RADIUS[scale] if radius is None else radius
    cpu = lambda t: t.detach().float().cpu()
    b1, r1, b2, r2 = cpu(build_taps["pass1_input"]), cpu(ref_taps["pass1_input"]), cpu(build_taps["pass2_input"]), cpu(ref_taps["pass2_input"])
    h, w = b1.shape[-2:]
    flips = lambda a, b: ((a[3] != b[3]) | (a[4] != b[4])).any(dim=0)          # [h,w]: any channel of either flow picture differs
    m2 = lambda t: (cpu(t).reshape(-1, h, w)[0] != 0)
    f1, f2, fm = flips(b1, r1), flips(b2, r2), m2(build_taps["vos_mask"]) != m2(ref_taps["vos_mask"])
    dil = lambda m: F.max_pool2d(m[None, None].float(), 2 * radius + 1, 1, radius)[0, 0] > 0
    excl = dil(f2 | fm | dil(f1))
    keep = (~excl).repeat_interleave(scale, 0).repeat_interleave(scale, 1).numpy()
    err = np.abs(got_hwc - ref_hwc)
    rng = np.abs(ref_hwc).max()
    out = dict(plane_flip_rate=dict(pass1_flow_pictures=round(float(f1.float().mean()), 6), pass2_flow_pictures=round(float(f2.float().mean()), 6),
                                    vos_mask=round(float(fm.float().mean()), 6)),
               flipped_pixels=dict(pass1_flow_pictures=int(f1.sum()), pass2_flow_pictures=int(f2.sum()), vos_mask=int(fm.sum()), of=int(h * w)),
               continuous_planes_max_rel_diff=dict(   # depth planes (5, 6) and the estimate plane (7) of pass 2: continuous, no flips
                   depth=float(f"{(b2[5:7] - r2[5:7]).abs().max().item() / max(r2[5:7].abs().max().item(), 1e-30):.3e}")),
               excluded_fraction=round(float(excl.float().mean()), 4), radius_lr_px=radius)
    if keep.any():
        e = err[keep]
        out.update(max_rel_err_outside=float(f"{e.max() / rng:.3e}"), p99_rel_err_outside=float(f"{np.percentile(e, 99) / rng:.3e}"),
                   psnr_outside_db=round(float(10 * np.log10(255.0 ** 2 / max(float(np.mean(e ** 2)), 1e-30))), 2))
    else:
        out.update(max_rel_err_outside=None, p99_rel_err_outside=None, psnr_outside_db=None)
    return out


def accuracy_vs_oracle(model, scale, precision, dev, first=None, lr=64):
    """PSNR (peak 255) and max error relative to the oracle's value range of ONE VSR.forward (no recurrent estimate) on an LR
    `lr` x `lr` tile of the synthetic clip: this build on the GPU against the oracle (the CPU restatement of the reference's
    forward, pinned by the reference-generated fixtures) on identical inputs and weights.  `first`: the
    oracle's (lr, data, output) if cpu_baseline already evaluated that tile."""
    if first is None:
        from oracle import vsr_oracle
        from video_super_resolution_amd import VSR
        from video_super_resolution_amd.weights import fill_module_
        torch.set_num_threads(max(1, min(16, host_cores())))
        torch.set_flush_denormal(True)
        P = {k: v.detach() for k, v in fill_module_(VSR(upscale_factor=scale).eval(), seed=0).state_dict().items()}
        data = torch.from_numpy(synthetic_clip(0, 3, lr, lr))
        otaps = {}
        with torch.no_grad():
            first = (lr, data, vsr_oracle.vsr_forward(P, data, None, upscale_factor=scale, taps=otaps), otaps)
    lr, data, ref, ref_taps = first
    hf = torch.zeros((3, scale * lr, scale * lr, 3), dtype=torch.float32, device=dev)
    model.plane_taps = {}
    try:
        with torch.no_grad():
            got, _ = model(data.to(dev), None, hf, None, train=False)
        build_taps = model.plane_taps
    finally:
        model.plane_taps = None
    got, ref = got.float().cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
    try:   # which part of the error is the discrete guidance planes
        flips = plane_flip_report(build_taps, ref_taps, got[0], ref[0], scale)
    except Exception as exc:   # noqa: BLE001  (an extra: must not cost the line)
        flips = dict(error=repr(exc)[:200])
    # the SR stack alone on IDENTICAL planes (no discrete guidance plane between this build and the oracle): what the arithmetic itself does
    sr_only = None
    try:
        from oracle import vsr_oracle
        rs = np.random.RandomState(1)
        planes = torch.from_numpy(rs.randint(0, 256, (8, 3, 24, 32)).astype(np.float32))
        Psr = {k[len("model."):]: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("model.")}
        with torch.no_grad():
            want = vsr_oracle.sr_forward(Psr, planes, upscale_factor=scale).numpy().astype(np.float64)
            have = model.model(planes.to(dev)).float().cpu().numpy().astype(np.float64)
        sr_only = float(f"{np.abs(have - want).max() / np.abs(want).max():.3e}")
    except Exception as exc:   # noqa: BLE001  (an extra: must not cost the line)
        sr_only = repr(exc)[:120]
    err = np.abs(got - ref)
    mse = float(np.mean(err ** 2))
    rng = np.abs(ref).max()
    return dict(psnr_vs_oracle_db=round(10 * np.log10(255.0 ** 2 / max(mse, 1e-30)), 2),
                max_rel_err=float(f"{err.max() / rng:.3e}"), p99_rel_err=float(f"{np.percentile(err, 99) / rng:.3e}"),
                sr_stack_max_rel_err_identical_planes=sr_only, guidance_plane_flips=flips,
                median_rel_err=float(f"{np.percentile(err, 50) / rng:.3e}"),
                note="the guidance planes are DISCRETE (uint8 flow pictures, the 0/1 VOS mask): a rounding-level difference in a trunk flips "
                     "a few plane pixels by a whole step, which is where the maximum comes from; the percentiles describe the frame; "
                     "sr_stack_max_rel_err_identical_planes = the SR net alone (this build vs the oracle) on identical 8 x 3 x 24 x 32 planes: the arithmetic itself",
                tile=f"one VSR.forward (estimated_image None) on LR {lr}x{lr} of the synthetic clip, x{scale}, {precision}: this build "
                     f"on the GPU vs the oracle on the CPU, same seeded weights; max_rel_err = max |diff| / max |oracle|",
                oracle_range=[round(float(ref.min()), 3), round(float(ref.max()), 3)])


def other_configs(progress, names=("C3B", "C5", "C2"), steps=5, warmup=2, timeout_s=170):
    """The other single-GPU configurations of BASELINE.json, each as a SHORT run of this same script in a child process
    (own model, own allocator; a failure there cannot cost the headline line), so that the driver's one default invocation
    also times reading B of the headline ("1080p LR -> 4K", C3-B), C5 and C2.  -> {name: {value, ms_per_step, roofline, ...}}"""
    import subprocess
    torch.cuda.empty_cache()
    out = {}
    for name in names:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", name, "--steps", str(steps), "--warmup", str(warmup),
               "--no-extras", "--no-cpu-baseline", "--no-configs", "--accuracy"]
        progress(f"config {name}: {steps} steps in a child process")
        t0 = time.time()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or len(lines) != 1:
                out[name] = dict(error=f"rc {p.returncode}: {p.stderr[-300:]}")
                continue
            d = json.loads(lines[0])
            out[name] = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline",
                                           "psnr_vs_oracle_db", "max_rel_err", "accuracy") if k in d}
            out[name]["wall_s"] = round(time.time() - t0, 1)
        except subprocess.TimeoutExpired:
            out[name] = dict(error=f"no line within {timeout_s} s")
    return out


def self_launch(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` as a CHILD process (never an exec;
    nothing in this process has touched the GPU yet), let its ranks print to this process's stdout / stderr -- rank 0 prints the one
    JSON line -- and return its exit code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: RCCL across processes needs it on this image)
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"[bench] --gpus {n_gpus} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="C3A", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: headline C3-A)")
    ap.add_argument("--lr-h", type=int, default=None)
    ap.add_argument("--lr-w", type=int, default=None)
    ap.add_argument("--scale", type=int, default=None, choices=[2, 3, 4])
    ap.add_argument("--precision", default=None, choices=["fp16", "fp32"],
                    help="SR stack: fp16 storage + fp32 accumulate on MFMA (headline config) or exact fp32")
    ap.add_argument("--clips", type=int, default=0,
                    help="config C4: this many independent clips round-robin over the ranks (clip i -> rank i mod N), `steps` "
                         "frames each, one asynchronous gather per finished clip; 0 = one clip per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accuracy", action="store_true",
                    help="with --no-cpu-baseline: still evaluate the oracle on ONE LR 64x64 tile (a few CPU seconds) for the line's "
                         "psnr_vs_oracle_db / max_rel_err fields (the default run takes them from the cpu_baseline's first tile)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and issue every collective (barrier, gathers, all_reduce) even with ONE "
                         "rank: lets a one-GPU box execute the N > 1 code path (tests/test_gpu_rccl_single_rank.py)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default C3-A line only: do not append the short C3-B / C5 / C2 runs under `configs`")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the separately-labelled extra loops (opt-in streaming mode; graph replay; PCIe-inclusive uint8 in / uint8 out)")
    args = ap.parse_args()

    h, w, scale, precision, label = CONFIGS[args.config]
    h, w = args.lr_h or h, args.lr_w or w
    scale = args.scale or scale
    precision = args.precision or precision

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: become one (before anything touches the GPU), forward the ranks' output and exit code
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    n_vis = torch.cuda.device_count()   # (counting devices does not initialise the GPU on this image)
    if args.gpus > n_vis:   # (every rank says so and leaves before any of them has touched a GPU or opened the process group)
        raise SystemExit(f"[bench rank {rank}] --gpus {args.gpus} needs {args.gpus} visible GPUs (one process per GPU, RCCL over xGMI); this box has "
                         f"{n_vis}")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)

    from video_super_resolution_amd import VSR, _lib
    from video_super_resolution_amd.distributed import clips_of_rank, gather_frames, run_sharded_clips
    from video_super_resolution_amd.weights import fill_module_

    model = fill_module_(VSR(upscale_factor=scale).eval(), seed=0).to(dev)
    model.precision = model.model.precision = precision
    H, W = scale * h, scale * w
    n_clips = args.clips if args.clips > 0 else world
    if n_clips < world:
        raise SystemExit("--clips must be at least the number of ranks")
    my_clips = clips_of_rank(n_clips, rank, world)
    n_frames = args.steps + args.warmup + 2
    # every clip of this rank resident in HBM before timing (a [K+W+2, h, w, 3] f32 clip is 50 MB at 540x960)
    clips = {c: torch.from_numpy(synthetic_clip(c, n_frames, h, w)).to(dev) for c in my_clips}
    hf = torch.zeros((3, H, W, 3), dtype=torch.float32, device=dev)

    def progress(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    progress(f"{label}, {precision}; model and {len(my_clips)} clip(s) resident on {torch.cuda.get_device_name(dev)}; warm-up")
    first = clips[my_clips[0]]
    # the dominant kernel's launches, timed under one name per plane count so that a launch is always priced by the planes it
    # processed: both SR passes launch it on the 5 planes they do not share (the 3 LR-frame planes run once per forward, on a
    # side stream beside the guidance trunks: VSR.overlap_shared); 8-plane launches only with plane sharing switched off
    # (x4: with VSR.early_planes the passes run on 4 planes, plane 7 of each pass on a side stream: `_side` marks launches that share the chip
    # with the guidance trunks)
    dom_names = ({"sr_utd_f16"} | {f"sr_utd_f16_p{n}{sd}" for n in range(1, 8) for sd in ("", "_side")}) if (precision == "fp16" and scale == 4) else \
        (({"sr_utd_s2_f16", "sr_stage_up", "sr_stage_dt", "sr_stage_dn"} | {f"sr_utd_s2_f16_p{n}{sd}" for n in range(1, 8) for sd in ("", "_side")}) if precision == "fp16" else
         {f"{k}{p}" for k in ("sr_conv8s4_f32", "sr_deconv8s4_f32", "sr_conv_f32", "sr_deconv_f32", "sr_deconv_dt_f32") for p in ("", "_p5", "_p3")})
    with torch.no_grad():
        # initialisation, not a step: both entry paths of forward (no estimate yet / recurrent estimate) run once so that
        # weight packing, executor construction and the caching allocator's first-touch hipMallocs are outside the clock
        prime = None
        for _ in range(2):
            prime, _ = model(first[0:3], None, hf, prime, train=False)
        del prime
        torch.cuda.synchronize()
        est = None
        for t in range(args.warmup):
            est, _ = model(first[t:t + 3], None, hf, est, train=False)
            torch.cuda.synchronize()
            progress(f"warm-up frame {t} done")
        warm_est = est

        def forward_clip(cid):
            """`steps` recurrent frames of clip `cid` -> [K, H, W, 3] fp16 (pixel values 0..255-ish)."""
            clip = clips[cid]
            out = torch.empty((args.steps, H, W, 3), dtype=torch.float16, device=dev)
            e = warm_est if cid == my_clips[0] else None   # the first clip continues from its warm-up frames
            for i, t in enumerate(range(args.warmup, args.warmup + args.steps)):
                e, _ = model(clip[t:t + 3], None, hf, e, train=False)
                out[i].copy_(e[0])
            return out

        if use_dist:
            dist.barrier()
        # HIP events around the dominant kernel's launches only: ~12 event pairs per frame, not one per launch
        _lib.TIMER.reset()
        _lib.TIMER.only = dom_names
        _lib.TIMER.enabled = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gather_info = None
        if args.clips > 0:
            gathered, ran = run_sharded_clips(forward_clip, n_clips, rank, world, dst=0, force_collective=args.force_dist)   # [n_clips, K, H, W, 3] on rank 0
            total_frames = n_clips * args.steps
            gather_info = dict(getattr(run_sharded_clips, "last", {}))
        else:
            finished = forward_clip(my_clips[0])
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            g = gather_frames(finished, dst=0, force_collective=args.force_dist)
            ev[1].record()
            gathered = torch.stack(g) if g is not None else None
            ran = 1
            total_frames = world * args.steps
        torch.cuda.synchronize()
        own_s = time.perf_counter() - t0       # this rank's own clock: its clips + its part of the gather(s), before the barrier
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if gather_info is None:
            gather_info = dict(gather_wait_ms=[round(ev[0].elapsed_time(ev[1]), 3)], rounds=1,
                               root_resident_bytes=int(sum(t.numel() * t.element_size() for t in g)) if g is not None else 0)
        _lib.TIMER.enabled = False
        _lib.TIMER.only = None
        progress(f"timed region: {elapsed:.3f} s for {args.steps} steps x {ran} clip(s) on this rank")
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    # per-rank decomposition for the first multi-GPU run (the value below stays frames / MAX-over-ranks time): every rank's own
    # seconds and frames, gathered outside the timed region
    mine = torch.tensor([own_s, float(ran * args.steps), float(sum(gather_info.get("gather_wait_ms", [])))], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(world)] if use_dist else [mine]
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, mine)
    elapsed = float(el.item())
    per_rank = [t.tolist() for t in per_rank]

    # ---- two extra loops, reported beside the headline value and never part of it (single GPU only)
    extras = {}
    def run_extras():
        from video_super_resolution_amd import driver
        clip = clips[my_clips[0]]
        with torch.no_grad():
            # (1) opt-in streaming mode: depth predictions / flow pictures of the frames consecutive windows share are kept
            model.temporal_cache = True
            e = warm_est
            for t in range(2):
                e, _ = model(clip[t:t + 3], None, hf, e, train=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for t in range(args.warmup, args.warmup + args.steps):
                e, _ = model(clip[t:t + 3], None, hf, e, train=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            model.temporal_cache = False
            model.reset_temporal_cache()
            extras["streaming"] = dict(value=round(args.steps / dt, 4), unit="frames/s", ms_per_step=round(1e3 * dt / args.steps, 3),
                                       note="VSR.temporal_cache = True: depth predictions and flow pictures of the two frames "
                                            "consecutive windows share are reused across calls (same arithmetic on smaller trunk batches; skips "
                                            "work the reference's per-window forward repeats) -- NOT the headline value")
            # (1b) opt-in graph replay (GraphedVSR): the same launches, issued by one hipGraphLaunch per frame instead of ~900 host calls
            from video_super_resolution_amd import GraphedVSR
            gm = GraphedVSR(model, clone_output=False)
            e = warm_est
            for t in range(2):
                e, _ = gm(clip[t:t + 3], None, hf, e, train=False)
            torch.cuda.synchronize()
            e = warm_est
            t1 = time.perf_counter()
            for t in range(args.warmup, args.warmup + args.steps):
                e, _ = gm(clip[t:t + 3], None, hf, e, train=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            t1 = time.perf_counter()   # host time to ISSUE the same frames eagerly (no synchronise inside): what the replay removes
            e2 = warm_est
            for t in range(args.warmup, args.warmup + args.steps):
                e2, _ = model(clip[t:t + 3], None, hf, e2, train=False)
            issue = time.perf_counter() - t1
            torch.cuda.synchronize()
            extras["graph_replay"] = dict(value=round(args.steps / dt, 4), unit="frames/s", ms_per_step=round(1e3 * dt / args.steps, 3),
                                          eager_host_issue_ms_per_step=round(1e3 * issue / args.steps, 3),
                                          bit_identical_to_eager=bool(torch.equal(e, e2)),
                                          note="GraphedVSR: VSR.forward captured once as a HIP graph and replayed (same kernels, same values; "
                                               "host-independent) -- NOT the headline value")
            del gm
            # (2) PCIe-inclusive: a uint8 HR window [3,H,W,3] comes from pinned host memory, is resized / converted on the device
            # (main.py:155-159), and the uint8 HR frame goes back to pinned host memory, all on the compute stream
            win_host = torch.randint(0, 256, (1, 3, H, W, 3), dtype=torch.uint8).pin_memory()
            out_host = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
            e = warm_est
            for rep_ in range(2):
                t1 = time.perf_counter()
                for t in range(args.steps):
                    win = win_host.to(dev, non_blocking=True)
                    data, _, _ = driver.ingest_item(win, scale, want_hr=False)
                    e, _ = model(data[0], None, hf, e, train=False)
                    out_host.copy_(driver.frames_to_u8(e[0]), non_blocking=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
            extras["pcie_inclusive"] = dict(value=round(args.steps / dt, 4), unit="frames/s", ms_per_step=round(1e3 * dt / args.steps, 3),
                                            h2d_bytes_per_step=int(win_host.numel()), d2h_bytes_per_step=int(out_host.numel()),
                                            note="uint8 HR window host -> device, nearest x1/scale + float on the device, forward, "
                                                 "uint8 HR frame device -> host; serial on one stream -- NOT the headline value")

    if world == 1 and args.clips == 0 and not args.no_extras:
        try:   # the extras must never cost the headline line
            run_extras()
        except Exception as exc:   # noqa: BLE001
            extras["extras_error"] = repr(exc)[:300]
    if rank == 0:
        assert gathered is not None and gathered.shape[0] * gathered.shape[1] == total_frames
        assert torch.isfinite(gathered[0].float()).all() and torch.isfinite(gathered[-1].float()).all()

    if rank == 0:
        fps = total_frames / elapsed
        calls_per_rank = ran * args.steps
        ms_per_frame = 1e3 * elapsed / calls_per_rank
        # ---- roofline of the dominant kernel, timed with HIP events inside the timed region
        timers = _lib.TIMER.summary()
        # the fused stage's launches by plane count: main-stream launches (the two SR passes: 5 planes, 4 with VSR.early_planes) and
        # side-stream ones (`_side`; x2: the 3-plane launches), which run BESIDE the guidance trunks
        utd_base = "sr_utd_f16" if scale == 4 else "sr_utd_s2_f16"
        mains, sides = {}, {}
        for k in [k for k in timers if k == utd_base or k.startswith(utd_base + "_p")]:
            v = timers.pop(k)
            kk = k[:-5] if k.endswith("_side") else k
            pl = int(kk.rsplit("_p", 1)[1]) if kk != utd_base else 8
            (sides if k.endswith("_side") else mains)[pl] = v
        part, part3, planes_dom = None, None, 8
        if mains:   # the dominant kernel's launches = the main-stream group with the most time
            planes_dom, v = max(mains.items(), key=lambda kv: kv[1][0] * kv[1][1])
            timers[utd_base] = v
            part3 = sides.get(3)
        dom = max(timers.items(), key=lambda kv: kv[1][0] * kv[1][1]) if timers else None
        roof = None
        if dom is not None:
            name, (launches, ms) = dom
            traffic, traffic_source = None, None
            if name in ("sr_utd_f16", "sr_utd_s2_f16"):
                # algorithmic FLOPs per launch: the fused up -> tran -> down stage on the launch's planes
                flop = planes_dom * h * w * STAGE_FLOP_PER_PX[scale]
                achieved = flop / (ms * 1e-3) / 1e12
                # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
                # gfx950 correction applied) -- a committed measurement of the same launch geometry, not taken in this run
                # (file, planes of the profiled launch): the 5-plane flat-split launch has its own passes (round 3)
                pmcs = {(540, 960, 4): (("r05_k_utd4_post_pmc_4planes.json", 4), ("r05_k_utd3_pmc_4planes.json", 4), ("r05_k_utd3_pmc_5planes.json", 5), ("r03_k_utd3_pmc.json", 5), ("r02_k_utd3_pmc.json", 8)),
                        (1080, 1920, 2): (("r05_k_utd_s2_post_pmc_8planes.json", 8), ("r05_k_utd_s2_pmc_8planes.json", 8), ("r02_k_utd_s2_pmc.json", 8)),
                        (2160, 3840, 2): (("r04_c5_k_utd_s2_hbm_pmc.json", 5),)}
                cands = sorted(pmcs.get((h, w, scale), ()), key=lambda fp: fp[1] != planes_dom)   # the launch's own geometry first
                for pmc, pl in cands:
                    path = os.path.join(ROOT, "profiles", pmc)
                    if os.path.exists(path):
                        with open(path) as f:
                            traffic = json.load(f)["hbm"]["traffic_bytes_per_launch"] * planes_dom / float(pl)
                        traffic_source = f"profiles/{pmc} (rocprofv3 --pmc, separate run of the {pl}-plane launch geometry" + \
                            (")" if planes_dom == pl else f", scaled to {planes_dom} planes: the traffic is per plane)")
                        if "_post_" in pmc:
                            # the profiled launch carries the fused uptran 1x1 (a second [planes,h,w,32] fp16 output); every other stage launch
                            # of a pass does not: the timed launches are half and half
                            traffic -= 0.5 * planes_dom * h * w * 64
                            traffic_source += "; that launch writes the fused 1x1's second output, which half of the timed launches do not: minus half of it"
                        break
                roof = dict(bound="mfma", kernel=name, achieved=round(achieved, 3), peak=FP16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(achieved / FP16_MFMA_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source,
                            launches_timed=launches, avg_ms=round(ms, 4), algorithmic_flop_per_launch=flop, planes_per_launch=planes_dom)
                # what an fp16 MFMA loop of this kernel's instruction shape SUSTAINS on this chip (power-governed clock): a committed
                # same-device measurement (tools/power_roofline.py), reported beside the spec peak -- `frac` stays achieved / spec peak
                pr_path = os.path.join(ROOT, "profiles", "r05_power_roofline.json")
                if scale == 4 and os.path.exists(pr_path):
                    with open(pr_path) as f:
                        pr = json.load(f)
                    roof["practical_peak"] = dict(
                        value=pr["mix_loop_tflops"], unit="TFLOP/s", frac_of_practical=round(achieved / pr["mix_loop_tflops"], 4),
                        bare_mfma_loop_tflops=pr["bare_mfma_tflops"], source="profiles/r05_power_roofline.json (tools/power_roofline.py: 144 "
                        "v_mfma_f32_16x16x32_f16 + k_utd3's 304 VALU + 17 LDS per trip, no global memory, random operands, one wave per SIMD, "
                        ">= 2.5 s back to back with k_utd3 on one device; measured on another box than this run)")
                if part3 is not None:
                    f3 = 3 * h * w * STAGE_FLOP_PER_PX[scale]
                    roof["three_plane_launches"] = dict(launches_timed=part3[0], avg_ms=round(part3[1], 4), algorithmic_flop_per_launch=f3,
                                                        achieved=round(f3 / (part3[1] * 1e-3) / 1e12, 3),
                                                        note="the LR-frame planes, once per forward on a side stream BESIDE the guidance trunks "
                                                             "(they share the chip: not a clean kernel time)")
                others = [(pl, v, False) for pl, v in sorted(mains.items()) if pl != planes_dom] + [(pl, v, True) for pl, v in sorted(sides.items()) if pl != 3]
                if others:
                    roof["other_launches"] = [dict(planes=pl, side_stream=sd, launches_timed=v[0], avg_ms=round(v[1], 4),
                                                   achieved=round(pl * h * w * STAGE_FLOP_PER_PX[scale] / (v[1] * 1e-3) / 1e12, 3)) for pl, v, sd in others]
            elif name.startswith("sr_stage_"):
                # unfused x2 stage (scale extension): each of its three launches is an HBM pass over the HR map.
                # algorithmic bytes per LR pixel and plane (fp16, 32 ch = 64 B per pixel): up 64 in + 64 s^2 out;
                # dt 64 s^2 in + 64 s^2 out; dn 64 s^2 in + 64 out.  A launch covers `planes` planes (chunked below 2 GiB).
                from video_super_resolution_amd.sr import _planes_per_chunk
                planes = _planes_per_chunk(8, scale * h, scale * w)
                per_px = {"sr_stage_up": 64 + 64 * scale ** 2, "sr_stage_dt": 128 * scale ** 2, "sr_stage_dn": 64 * scale ** 2 + 64}[name]
                nbytes = planes * h * w * per_px
                achieved = nbytes / (ms * 1e-3) / 1e9
                roof = dict(bound="hbm", kernel=name, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None, traffic_source=None, launches_timed=launches,
                            avg_ms=round(ms, 4), algorithmic_bytes_per_launch=nbytes, planes_per_launch=planes)
            else:
                planes = int(name.rsplit("_p", 1)[1]) if "_p" in name[-3:] else 8    # (the second SR pass runs on the 5 planes the passes do not share)
                name = name[:-3] if "_p" in name[-3:] else name
                flop = planes * h * w * (STAGE_FLOP_PER_PX[scale] - 2048 * scale ** 2) / 2    # one k x k (de)conv
                if name == "sr_deconv_dt_f32":
                    flop += planes * h * w * 2048 * scale ** 2    # + the downtran 1x1 (32 x 32 MACs per HR pixel) in its epilogue
                achieved = flop / (ms * 1e-3) / 1e12
                traffic, traffic_source = None, None
                pmc = os.path.join(ROOT, "profiles", "r04_c2_deconv_f32_hbm_pmc.json")
                if name == "sr_deconv_dt_f32":
                    pmc = os.path.join(ROOT, "profiles", "r04_c2_deconv_dt_f32_hbm_pmc.json")
                if name in ("sr_deconv_f32", "sr_deconv_dt_f32") and (h, w, scale) == (540, 960, 2) and os.path.exists(pmc):
                    with open(pmc) as f:
                        traffic = json.load(f)["hbm"]["traffic_bytes_per_launch"] * planes / 8.0
                    traffic_source = f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate run of the 8-plane launch" + \
                        (")" if planes == 8 else f", scaled to {planes} planes: the traffic is per plane)")
                roof = dict(bound="mfma", kernel=name, achieved=round(achieved, 3), peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(achieved / FP32_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source, launches_timed=launches,
                            avg_ms=round(ms, 4), algorithmic_flop_per_launch=flop, planes_per_launch=planes)
                # what a bare v_mfma_f32_32x32x2_f32 loop sustains on this chip (a committed same-device measurement, tools/power_roofline_f32.py):
                # unlike the fp16 MFMA the float32 one runs at the full clock, so the spec figure IS reachable and the gap is the kernel's own
                pp = os.path.join(ROOT, "profiles", "r04_power_roofline_f32.json")
                if os.path.exists(pp):
                    with open(pp) as f:
                        pj = json.load(f)
                    roof["practical_peak"] = dict(value=pj["bare_mfma_f32_tflops"], unit="TFLOP/s", frac_of_practical=round(achieved / pj["bare_mfma_f32_tflops"], 4),
                                                  source="profiles/r04_power_roofline_f32.json (tools/power_roofline_f32.py: bare v_mfma_f32_32x32x2_f32 loop, operands in "
                                                         "registers, one wave per SIMD, 2.5 s back to back with the block kernels on one device: 2.38 GHz, 64.0 cycles per MFMA)")
            # the whole frame against the same peak: FLOPs this implementation executes per forward / wall time per forward
            peak = FP16_MFMA_PEAK_TFLOPS if precision == "fp16" else FP32_PEAK_TFLOPS
            exe = executed_flop_per_frame(h, w, scale, precision=precision)
            roof["whole_frame"] = dict(executed_flop_per_frame=exe, achieved_tflops=round(exe / (ms_per_frame * 1e-3) / 1e12, 2),
                                       frac_of_peak=round(exe / (ms_per_frame * 1e-3) / 1e12 / peak, 4), peak_tflops=peak)
        line = dict(metric=f"HR frames/sec, 1080p->4K x4 VSR (LR {h}x{w} -> {H}x{W}), VSR.forward end-to-end" if (args.config, h, w, scale) == ("C3A", 540, 960, 4)
                    else f"HR frames/sec, VSR.forward end-to-end (LR {h}x{w} x{scale} -> {H}x{W})",
                    value=round(fps, 4), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(ms_per_frame, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32" if precision == "fp32" else "f16", data="synthetic",
                    config=dict(workload=f"{label if (h, w, scale) == CONFIGS[args.config][:3] else f'LR {h}x{w} x{scale}'}, 3-frame window "
                                         f"+ recurrent estimate, {n_clips} clip(s) over {world} GPU(s), seeded synthetic weights",
                                precision=precision, parallelism=f"clip-dp{world}", clips=n_clips, scale=scale),
                    roofline=roof)
        line.update(extras)
        line["env"] = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "MIOPEN_FIND_MODE")}   # effective values
        if world == 1 and not args.no_cpu_baseline:
            progress("timing the CPU oracle on LR 64x64 / 96x96 / 128x128 tiles (about a minute)")
            torch.cuda.synchronize()
            cb = cpu_baseline(scale)
            sec_full = cb["intercept_s"] + cb["slope_s_per_px"] * h * w
            pts = ", ".join(f"{int(p ** 0.5)}x{int(p ** 0.5)}: {s} s" for p, s in cb["points"])
            acc = accuracy_vs_oracle(model, scale, precision, dev, first=cb["first"])
            line["psnr_vs_oracle_db"], line["max_rel_err"], line["accuracy"] = acc["psnr_vs_oracle_db"], acc["max_rel_err"], acc
            line["cpu_baseline"] = dict(value=round(1.0 / sec_full, 8), unit="frames/s", cores=cb["cores"], kind="port",
                                        cores_available=cb["cores_available"], cpu=cb["cpu"], points=cb["points"],
                                        sample=f"1 frame of VSR.forward (the oracle, x{scale}) at LR {pts}; least-squares line "
                                               f"{cb['intercept_s']:.2f} s + {cb['slope_s_per_px'] * 1e3:.4f} ms/px evaluated at "
                                               f"LR {h}x{w} = {sec_full:.0f} s per frame (extrapolated)")
        elif world == 1 and args.accuracy:
            acc = accuracy_vs_oracle(model, scale, precision, dev)
            line["psnr_vs_oracle_db"], line["max_rel_err"], line["accuracy"] = acc["psnr_vs_oracle_db"], acc["max_rel_err"], acc
        if (world == 1 and args.config == "C3A" and not args.no_configs and not args.force_dist and args.clips == 0 and
                args.lr_h is None and args.lr_w is None and args.scale is None and args.precision is None):
            line["configs"] = other_configs(progress)
        line["ranks"] = dict(frames_per_s=[round(f / t, 3) for t, f, _ in per_rank], own_seconds=[round(t, 4) for t, _, _ in per_rank],
                             gather_wait_ms=[round(gw, 3) for _, _, gw in per_rank],
                             note="per rank: frames it produced / its own wall time up to the end of its gather(s) (before the closing "
                                  "barrier); gather_wait_ms = time its stream stood behind the gather transfers")
        line["gather"] = dict(rounds=gather_info.get("rounds"), root_resident_bytes=gather_info.get("root_resident_bytes"),
                              root_wait_ms_per_round=gather_info.get("gather_wait_ms"),
                              note="rank 0: receive buffers it holds for the gathered frames (C4: 32 clips x K frames x 49.8 MB fp16) and "
                                   "how long each round's transfer kept its stream waiting")
        if use_dist:
            line["process_group"] = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), forced_single_rank=bool(args.force_dist))
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
