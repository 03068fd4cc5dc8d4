"""Clip-level data parallelism: the only multi-GPU structure the path admits.

Frames inside a clip are serial (`estimated_image` recurrence, main.py:196-202) and the second SR pass of a
frame depends on the first, but clips share no state, so clip i runs on rank i mod W with a full weight replica
and NO collective on the data path.  The single exchange step is the gather of finished HR frames to rank 0 at
the end (one `torch.distributed.gather`; backend "nccl" is RCCL on ROCm: W-1 direct xGMI transfers into the
root, not a ring).  The reference itself has no multi-GPU code at all (main.py:26,60-63).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def clips_of_rank(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: clip i -> rank i mod world."""
    return list(range(rank, n_clips, world))


def gather_frames(local: torch.Tensor, dst: int = 0, group=None, force_collective: bool = False) -> Optional[List[torch.Tensor]]:
    """Gather every rank's finished frames `[k_r, ...]` on `dst`; ranks may hold different k_r.

    Returns the list of per-rank tensors (trimmed to their true length) on `dst`, None elsewhere.
    `force_collective`: issue the collectives even in a process group of ONE rank (a one-GPU box can then execute the RCCL
    path -- communicator init, device-side gather buffers, stream ordering -- that otherwise first runs on an 8-GPU node).
    """
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    kmax = max(counts)
    if local.shape[0] < kmax:  # pad to a common shape: gather needs equal sizes
        pad = torch.zeros((kmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat((local, pad), 0)
    local = local.contiguous()
    # `dst` is a GLOBAL rank (what dist.gather takes); `rank` is this process's rank inside `group`: compare like with like
    dst_in_group = dist.get_group_rank(group, dst) if group is not None else dst
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst_in_group else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst_in_group:
        return None
    return [b[:c] for b, c in zip(bufs, counts)]


def interleave_clips(per_rank: List[torch.Tensor], n_clips: int) -> torch.Tensor:
    """Undo `clips_of_rank`: per-rank stacks `[k_r, ...]` -> `[n_clips, ...]` in clip order."""
    world = len(per_rank)
    out = [None] * n_clips
    for r, t in enumerate(per_rank):
        for j, cid in enumerate(range(r, n_clips, world)):
            out[cid] = t[j]
    return torch.stack(out)


class ClipGather:
    """The gather of config C4 issued clip by clip: as soon as a rank has finished a clip, its frames leave for `dst`
    (asynchronous `gather`, one per round of W clips) while the rank computes its next clip -- the xGMI transfers
    (7 direct links into the root) overlap with compute instead of forming one end-of-job burst of `clips_per_rank`
    times the size.  Every rank calls `submit` once per round, in the same order (a collective); a rank without a clip
    in the last round (n_clips not a multiple of W) submits None.  `finish()` waits for all rounds and returns
    `[n_clips, K, ...]` in clip order on `dst`, None elsewhere.  Single process (no process group): pass-through."""

    def __init__(self, n_clips: int, dst: int = 0, group=None, force_collective: bool = False):
        """`force_collective`: see gather_frames.  `dst` is a GLOBAL rank (what dist.gather takes); with a sub-group it is translated to the group rank for the
        "am I the root" test."""
        self.n_clips, self.dst, self.group = n_clips, dst, group
        self.dist = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank = dist.get_rank(group) if self.dist else 0
        # group rank of the root; single process: whatever `dst` was asked for, this process is the root
        self.dst_in_group = (dist.get_group_rank(group, dst) if group is not None else dst) if self.dist else 0
        self.rounds = []   # (work handle or None, receive buffers or [local])
        self._shape = None
        self.wait_ms: List[float] = []   # per round: how long finish() stood waiting for that round's transfer (diagnostic)
        self.root_bytes = 0              # bytes of receive buffers this rank holds as the root (C4: 32 clips x K frames on rank 0)

    def submit(self, frames: Optional[torch.Tensor], like: Optional[torch.Tensor] = None):
        """frames [K, ...] of this rank's clip of the current round (None: no clip this round; `like` gives the shape)."""
        if frames is None:
            if like is None:
                raise ValueError("a rank without a clip must pass `like` (shape/dtype of a clip's frames)")
            frames = torch.zeros_like(like)
        frames = frames.contiguous()
        if not self.dist:
            self.root_bytes += frames.numel() * frames.element_size()
            self.rounds.append((None, [frames]))
            return
        bufs = [torch.empty_like(frames) for _ in range(self.world)] if self.rank == self.dst_in_group else None
        if bufs is not None:
            self.root_bytes += sum(b.numel() * b.element_size() for b in bufs)
        work = dist.gather(frames, bufs, dst=self.dst, group=self.group, async_op=True)
        self.rounds.append((work, bufs, frames))   # keep `frames` alive until the transfer has completed

    def finish(self) -> Optional[torch.Tensor]:
        import time
        out = []
        marks = []
        for entry in self.rounds:
            if entry[0] is not None:
                # RCCL: wait() orders the CURRENT STREAM behind the transfer (the host does not block), so the stall is
                # measured with events on that stream; gloo / CPU tensors: wait() blocks the host, measured on its clock
                on_gpu = entry[2].is_cuda
                if on_gpu:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    entry[0].wait()
                    e1.record()
                    marks.append((e0, e1))
                else:
                    t0 = time.perf_counter()
                    entry[0].wait()
                    marks.append(1e3 * (time.perf_counter() - t0))
            if self.rank == self.dst_in_group:
                out.extend(entry[1])
        self.wait_ms = []
        for m in marks:
            if isinstance(m, tuple):
                m[1].synchronize()
                self.wait_ms.append(m[0].elapsed_time(m[1]))
            else:
                self.wait_ms.append(m)
        self.rounds = []
        if self.rank != self.dst_in_group:
            return None
        # round j delivered clips j*W + r for r = 0..W-1, i.e. already in clip order; drop the padding of the last round
        return torch.stack(out[:self.n_clips])


def run_sharded_clips(forward_clip, n_clips: int, rank: int, world: int, dst: int = 0, group=None, force_collective: bool = False):
    """Control flow of config C4 (`bench.py --clips N`): clip i runs on rank i mod W, clips of a rank one after the other
    (frames inside a clip are serial), each finished clip gathered to `dst` while the next one runs.
    `forward_clip(clip_id) -> [K, ...]` produces the finished frames of one clip on this rank's device.
    Returns ([n_clips, K, ...] on dst | None, number of clips this rank ran)."""
    if n_clips < world:   # checked on EVERY rank BEFORE any collective: a clipless rank would leave the others' async
        #                   gathers waiting for it until the RCCL timeout
        raise RuntimeError(f"a rank without any clip cannot take part in the gather: n_clips ({n_clips}) must be >= the "
                           f"world size ({world})")
    mine = clips_of_rank(n_clips, rank, world)
    rounds = -(-n_clips // world)
    g = ClipGather(n_clips, dst=dst, group=group, force_collective=force_collective)
    like = None
    pending_none = 0
    for j in range(rounds):
        if j < len(mine):
            frames = forward_clip(mine[j])
            like = frames
            for _ in range(pending_none):   # (cannot happen with round-robin: a rank's missing clip is always its last)
                g.submit(None, like=like)
            pending_none = 0
            g.submit(frames)
        elif like is not None:
            g.submit(None, like=like)
        else:
            pending_none += 1
    assert pending_none == 0
    full = g.finish()
    run_sharded_clips.last = dict(gather_wait_ms=[round(x, 3) for x in g.wait_ms], root_resident_bytes=int(g.root_bytes), rounds=rounds)
    return full, len(mine)
