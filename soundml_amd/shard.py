"""Clip sharding across the GPUs of one node.

The path shards embarrassingly: leading axes (clips / channels) are independent
by contract (stft.mli:216-218, tested per slice in stft_grid.ml:180-205), so each
rank owns a contiguous clip range and NO collective is on the data path.  A rank
is one process bound to one GPU (``torch.distributed``; backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) -- the only communication is the timing barrier
of the benchmark and an optional host-side gather of results.
"""
from __future__ import annotations

from typing import Tuple


def clip_range(total_clips: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: rank r owns clips [lo, hi); the first
    ``total % world`` ranks own one extra clip.  Every clip is owned exactly once."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("clip_range: rank %d outside world of %d" % (rank, world_size))
    if total_clips < 0:
        raise ValueError("clip_range: negative clip count")
    base, extra = divmod(total_clips, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def frame_range(total_frames: int, parts: int, part: int) -> Tuple[int, int]:
    """Frame-range partition of ONE long clip for ``Stft.transform_range`` /
    ``power_range`` (stft.mli:226-240: adjacent ranges reassemble exactly)."""
    return clip_range(total_frames, parts, part)


def gather_host(local, group=None):
    """Optional result collection for callers that want one host tensor: every
    rank contributes its [clips_r; ...] block; returns the concatenation on every
    rank.  Not on the hot path (the benchmark keeps shards device-resident)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, local.cpu() if hasattr(local, "cpu") else local, group=group)
    return torch.cat([torch.as_tensor(p) for p in parts], dim=0)


def timed_region_max(elapsed_seconds: float, device=None, group=None) -> float:
    """The benchmark's clock: every rank times the same K steps between two barriers; the job's
    time is the MAX over ranks (one all-reduce of a scalar, outside the data path)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(elapsed_seconds)
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(group=None) -> None:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)
