"""Clip sharding + the single end-of-job gather, rehearsed with two gloo ranks on the CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    from video_super_resolution_amd.distributed import clips_of_rank, gather_frames, interleave_clips
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = clips_of_rank(n_clips, rank, world)
    # a "finished frame" of clip c is a small tensor filled with c (+ pixel index) so order is checkable
    local = torch.stack([torch.full((4, 6, 3), float(c)) + torch.arange(3.0) for c in mine]) if mine else torch.zeros(0, 4, 6, 3)
    got = gather_frames(local, dst=0)
    if rank == 0:
        full = interleave_clips(got, n_clips)
        q.put(full[:, 0, 0, 0].tolist())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5])
def test_two_rank_shard_and_gather(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res == [float(c) for c in range(n_clips)]  # every clip exactly once, in clip order


def test_round_robin_assignment_covers_every_clip_once():
    from video_super_resolution_amd.distributed import clips_of_rank
    for world in (1, 2, 4, 8):
        for n in (1, 7, 32):
            seen = sorted(c for r in range(world) for c in clips_of_rank(n, r, world))
            assert seen == list(range(n))
    assert clips_of_rank(32, 3, 8) == [3, 11, 19, 27]  # config 4: 32 clips, 4 per GPU


def _worker_clips(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    from video_super_resolution_amd.distributed import run_sharded_clips
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def stub_forward_clip(cid):   # stands in for K recurrent VSR.forward calls on this rank's device
        calls.append(cid)
        est = torch.zeros(2, 3)
        frames = []
        for t in range(3):       # K = 3 frames, each depending on the previous one (the estimated_image recurrence)
            est = est + cid + 0.1 * t
            frames.append(est.clone())
        return torch.stack(frames)

    full, n_mine = run_sharded_clips(stub_forward_clip, n_clips, rank, world, dst=0)
    assert calls == list(range(rank, n_clips, world)) and n_mine == len(calls)
    if rank == 0:
        q.put((tuple(full.shape), full[:, :, 0, 0].tolist()))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips", [(2, 4), (2, 5), (2, 2), (4, 6)])
def test_bench_clip_control_flow(world, n_clips):
    """`bench.py --clips N` (config C4) with a stub model: round-robin clips, per-clip asynchronous gathers, clip order
    restored on rank 0, ranks without a clip in the last round (4 ranks, 6 clips: the ragged last round of two clips)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_clips, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    shape, vals = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert shape == (n_clips, 3, 2, 3)
    for c in range(n_clips):
        expect = [c, 2 * c + 0.1, 3 * c + 0.3]
        assert vals[c] == pytest.approx(expect)


def test_clip_gather_single_process_is_a_pass_through():
    from video_super_resolution_amd.distributed import run_sharded_clips
    full, n = run_sharded_clips(lambda cid: torch.full((2, 4), float(cid)), 3, 0, 1)
    assert n == 3 and full.shape == (3, 2, 4) and full[:, 0, 0].tolist() == [0.0, 1.0, 2.0]


def _worker_subgroup(rank, world, port, q):
    """Ranks 1 and 2 of a 3-rank world form a sub-group; the root is GLOBAL rank 2 = group rank 1."""
    sys.path.insert(0, ROOT)
    from video_super_resolution_amd.distributed import ClipGather, gather_frames
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([1, 2])   # (every rank of the world calls new_group)
    if rank in (1, 2):
        local = torch.full((rank, 2, 3), float(rank))   # rank r holds r frames
        got = gather_frames(local, dst=2, group=grp)
        cg = ClipGather(2, dst=2, group=grp)
        cg.submit(torch.full((2, 3), 10.0 * rank))
        full = cg.finish()
        if rank == 2:
            q.put(([tuple(t.shape) for t in got], [float(t.flatten()[0]) for t in got], full[:, 0, 0].tolist()))
        else:
            assert got is None and full is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_root_given_as_global_rank_in_a_subgroup():
    """`dst` is a global rank; inside a sub-group the "am I the root" test must use the group rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    shapes, firsts, clip_vals = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert shapes == [(1, 2, 3), (2, 2, 3)] and firsts == [1.0, 2.0]
    assert clip_vals == [10.0, 20.0]


def test_bench_gpus_n_launches_itself_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: bench.py starts `python -m
    torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process before anything touches the GPU and exits
    with the child's code.  The spawn is replaced by a recorder here; under a launcher (RANK set) it must not spawn again."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = list(cmd), env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2", "--clips", "32"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--clips", "32"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # under a launcher whose world disagrees with --gpus: a clear message, no second spawn
    seen.clear()
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not seen and "WORLD_SIZE=1" in str(e.value.code)
