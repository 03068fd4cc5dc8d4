"""The RCCL code path executed on hardware with what a one-GPU box offers: a process group of ONE rank, backend "nccl"
(= RCCL on ROCm), every collective of the N > 1 path issued for real (`force_collective`) on device tensors.

It measures nothing -- one rank exchanges no bytes over xGMI -- but communicator initialisation, the device-side gather
buffers, asynchronous work handles, stream ordering between the compute stream and RCCL's, `all_reduce(MAX)` of the timing
scalar and the process-group teardown stop being code that first runs on the driver's 8-GPU node.
Each case runs in a FRESH process (the group is initialised before anything else touches the GPU there)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

_CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
import torch, torch.distributed as dist
from video_super_resolution_amd.distributed import ClipGather, gather_frames, run_sharded_clips
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
g = torch.Generator(device="cpu").manual_seed(3)
frames = torch.rand((3, 32, 48, 3), generator=g).to(dev).half()
# (1) the end-of-job gather of bench.py: all_gather of the counts + gather of the padded stacks, on device tensors
got = gather_frames(frames, dst=0, force_collective=True)
assert isinstance(got, list) and len(got) == 1 and torch.equal(got[0], frames)
# (2) config C4's per-clip asynchronous gathers, issued while "the next clip" is computed on the compute stream
def forward_clip(cid):
    x = frames * (cid + 1)
    for _ in range(4):
        x = x + 0          # some work on the compute stream between submits
    return x
out, ran = run_sharded_clips(forward_clip, 3, 0, 1, dst=0, force_collective=True)
assert ran == 3 and out.shape == (3,) + tuple(frames.shape)
for c in range(3):
    assert torch.equal(out[c], frames * (c + 1)), c
cg = ClipGather(2, dst=0, force_collective=True)
assert cg.dist and cg.world == 1
cg.submit(frames); cg.submit(None, like=frames)        # a rank without a clip in the last round submits zeros
fin = cg.finish()
assert fin.shape[0] == 2 and torch.equal(fin[0], frames) and float(fin[1].abs().max()) == 0.0
# (2b) the root given as a GLOBAL rank inside an explicit (sub-)group: group rank == global rank here, the code path is the same
grp = dist.new_group([0])
got = gather_frames(frames, dst=0, group=grp, force_collective=True)
assert isinstance(got, list) and len(got) == 1 and torch.equal(got[0], frames)
cg = ClipGather(1, dst=0, group=grp, force_collective=True)
cg.submit(frames)
assert torch.equal(cg.finish()[0], frames) and len(cg.wait_ms) == 1 and cg.root_bytes == frames.numel() * frames.element_size()
# (3) bench.py's max-over-ranks of the elapsed time
el = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(el, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(el.item()) == 1.25
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


def test_gather_helpers_on_a_one_rank_rccl_group():
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, str(_free_port())], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_SINGLE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.parametrize("extra", [[], ["--clips", "2"]])
def test_bench_under_the_distributed_launcher_with_one_rank(extra):
    """bench.py exactly as the driver launches N > 1 (`python -m torch.distributed.run --nproc-per-node N ...`), N = 1, with
    --force-dist: init_process_group("nccl"), barrier, the gathers, all_reduce(MAX), destroy -- all executed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--lr-h", "64",
           "--lr-w", "96", "--no-cpu-baseline", "--no-extras", "--force-dist"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["process_group"] == {"backend": "nccl", "world_size": 1, "forced_single_rank": True}
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert len(d["ranks"]["frames_per_s"]) == 1 and len(d["gather"]["root_wait_ms_per_round"]) == d["gather"]["rounds"]
    assert d["gather"]["root_resident_bytes"] >= 2 * 256 * 384 * 3 * 2     # K = 2 fp16 frames of 256 x 384 x 3 per clip


def test_bench_force_dist_through_the_plain_entry():
    """`python bench.py --gpus 1 --force-dist` (no launcher): the same entry the driver uses for N = 1 opens a one-rank RCCL group."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--lr-h", "64", "--lr-w", "96",
           "--no-cpu-baseline", "--no-extras", "--force-dist"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(_free_port())
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["process_group"]["backend"] == "nccl" and d["n_gpus"] == 1 and d["env"]["GPU_MAX_HW_QUEUES"] == "4"


def test_bench_gpus_2_on_a_one_gpu_box_starts_the_launcher_and_fails_with_the_ranks_message():
    """`python bench.py --gpus 2` by itself (how the driver calls N = 1): bench.py becomes the launcher -- a child
    `torch.distributed.run` with two ranks -- and on a box with fewer GPUs every rank leaves with a device-count message (not a usage
    error of the parent), before any of them has touched the GPU; the parent forwards the exit code."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--lr-h", "64", "--lr-w", "96",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "torch.distributed.run" in p.stderr and "needs 2 visible GPUs" in p.stderr, p.stderr[-3000:]
    assert "launch N>1 with" not in p.stderr
