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


def unpack_label_masks(masks, stride=None):
    """[N, S/4] uint8 2-bit masks (gg_batch.d_label_masks) -> [N, S] uint8 labels 0 / 49 / 99.  Pure torch (any device)."""
    import torch

    shifts = torch.tensor([0, 2, 4, 6], dtype=torch.uint8, device=masks.device)
    codes = (masks.unsqueeze(-1) >> shifts) & 3
    lut = torch.tensor([0, 49, 99, 0], dtype=torch.uint8, device=masks.device)
    out = lut[codes.long()].reshape(masks.shape[0], -1)
    return out if stride is None else out[:, :stride]


def pack_label_masks(labels):
    """Inverse of unpack_label_masks, pure torch (CPU tests; on the GPU the masks come straight out of k_label)."""
    import torch

    n, s = labels.shape
    assert s % 4 == 0
    codes = ((labels == 49).to(torch.uint8) + 2 * (labels == 99).to(torch.uint8)).reshape(n, s // 4, 4)
    return codes[..., 0] | (codes[..., 1] << 2) | (codes[..., 2] << 4) | (codes[..., 3] << 6)


def common_stride(local_max_points: int, group=None, device=None, multiple: int = 64) -> int:
    """Per-cloud buffer length every rank agrees on: the largest cloud anywhere, rounded up (the gathered masks need one
    shape on all ranks, and the 2-bit masks a multiple of 4).  One tiny all-reduce(MAX)."""
    import torch
    import torch.distributed as dist

    stride = (int(local_max_points) + multiple - 1) // multiple * multiple
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([stride], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        stride = int(t.item())
    return stride
