"""Multi-GPU driver pieces: frames are independent, so they are sharded over ranks with no data-path
collective; the only exchange is one all_gather of the per-frame road-width records (104 B each).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
This replaces the reference's strictly serial driver loop (semantic_depth_cityscapes_sequence.py:689-701).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 104      # sizeof(sd_rw_result)


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous block [lo, hi) of the frame list owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local: torch.Tensor, n_frames: int | None = None, group=None) -> torch.Tensor:
    """all_gather of the per-frame record buffers.  ``local``: uint8 [B_local, 104] on the rank's device.
    Returns uint8 [n_frames, 104] in global frame order on every rank (shards may be ragged: padded to the
    largest shard for the collective, then trimmed)."""
    assert local.dtype == torch.uint8 and local.dim() == 2 and local.shape[1] == RECORD_BYTES
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if n_frames is None:
        n_frames = local.shape[0] * world
    sizes = [shard_range(n_frames, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((bmax, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * bmax, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def records_view(buf: torch.Tensor) -> np.ndarray:
    from .engine import RW_DTYPE
    return buf.cpu().numpy().view(RW_DTYPE).reshape(-1)
