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


class AbiLabelGather:
    """The all-gather of the label masks through the library's own C entry point (gg_allgather_label_masks: RCCL bound with
    dlopen, no torch in the data path) -- what a C++ host of BASELINE configs[2] would call.  torch.distributed is used ONCE,
    to hand rank 0's 128-byte RCCL id to the other ranks (any transport would do); with world size 1 not at all."""

    def __init__(self, seg, rank: int = 0, world: int = 1, group=None):
        import ctypes as C

        from . import _lib

        self._L, self._seg, self.world, self.rank = _lib.load(), seg, world, rank
        if not self._L.gg_collective_available():
            raise _lib.GroundGridError("librccl.so could not be loaded")
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            rc = self._L.gg_comm_unique_id(ident)
            if rc != _lib.GG_OK:
                raise _lib.GroundGridError(f"gg_comm_unique_id: {_lib.STATUS.get(rc, rc)}")
        if world > 1:
            import torch
            import torch.distributed as dist

            t = torch.tensor(list(ident), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        comm = C.c_void_p()
        # (on the CONTEXT's device: with one rank or a gloo group nothing else has selected a HIP device in this process)
        rc = self._L.gg_comm_init_rank_for(seg._ctx, ident, world, rank, C.byref(comm))
        if rc != _lib.GG_OK:
            raise _lib.GroundGridError(f"gg_comm_init_rank_for: {_lib.STATUS.get(rc, rc)}")
        self._comm = comm

    def gather(self, masks, out=None, stream=None):
        """masks: contiguous CUDA uint8 tensor of this rank; returns [world * masks.shape[0], ...] on every rank.  Enqueued on
        the current torch stream (or `stream`), ordered after the context's last batch by the library."""
        import ctypes as C

        import torch

        from . import _lib

        assert masks.is_cuda and masks.is_contiguous() and masks.dtype == torch.uint8
        if out is None:
            out = torch.empty((self.world * masks.shape[0],) + tuple(masks.shape[1:]), dtype=torch.uint8, device=masks.device)
        s = stream if stream is not None else torch.cuda.current_stream(masks.device).cuda_stream
        rc = self._L.gg_allgather_label_masks(self._seg._ctx, self._comm, masks.data_ptr(), out.data_ptr(), masks.numel(),
                                              C.c_void_p(s if s else _lib.GG_STREAM_DEFAULT))
        if rc != _lib.GG_OK:
            raise _lib.GroundGridError(f"gg_allgather_label_masks: {_lib.STATUS.get(rc, rc)} {self._L.gg_last_error(self._seg._ctx).decode()}")
        return out

    def close(self):
        if getattr(self, "_comm", None):
            self._L.gg_comm_destroy(self._comm)
            self._comm = None
