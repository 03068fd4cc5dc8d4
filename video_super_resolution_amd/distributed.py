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


def gather_frames(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather every rank's finished frames `[k_r, ...]` on `dst`; ranks may hold different k_r.

    Returns the list of per-rank tensors (trimmed to their true length) on `dst`, None elsewhere.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
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
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
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
