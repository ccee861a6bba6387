"""Multi-GPU sharding of independent clouds (one process per GPU, torch.distributed over RCCL).

The path does not shard inside one stream of clouds (cloud k+1 reads the ground / groundpatch state cloud k
left, src/GroundSegmentation.cpp:243-275, :376-393); independent (cloud, map-state) pairs shard trivially with
no data-path exchange.  The only collective is one all-gather of the fixed-size per-cloud label masks per
batch (BASELINE.json configs[2]): ~N_max bytes per cloud, latency-bound over xGMI.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_clouds: int, rank: int, world: int) -> Tuple[int, int]:
    """Cloud b of n_clouds lives on rank b // per_rank, slot b % per_rank.  Returns (first, count)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = (n_clouds + world - 1) // world
    first = min(rank * per, n_clouds)
    return first, max(0, min(per, n_clouds - first))


def owner_of(cloud: int, n_clouds: int, world: int) -> Tuple[int, int]:
    per = (n_clouds + world - 1) // world
    return cloud // per, cloud % per


def all_gather_label_masks(labels, counts=None, group=None):
    """labels: [B, stride] uint8 on this rank (padded to the common stride).  Returns [world * B, stride]
    (and the gathered [world * B, 4] counts if given).  One collective; backend nccl (== RCCL) on GPU, gloo on CPU."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    B, stride = labels.shape
    out = torch.empty((world * B, stride), dtype=labels.dtype, device=labels.device)
    try:
        dist.all_gather_into_tensor(out, labels.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # older gloo builds
        parts = [torch.empty_like(labels) for _ in range(world)]
        dist.all_gather(parts, labels.contiguous(), group=group)
        out = torch.cat(parts, dim=0)
    if counts is None:
        return out
    cparts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(cparts, counts.contiguous(), group=group)
    return out, torch.cat(cparts, dim=0)
