"""Slice sharding across the GPUs of one node: one process per GPU, no collective inside the loop, ONE
gather at the end (RCCL over xGMI when the backend is 'nccl').

Slices are independent (GroupNorm and attention are per sample, t is shared), so the path shards as a pure map:
rank r owns the contiguous block [r*N/P, (r+1)*N/P) of the N slices, walks it in chunks of `chunk` slices and
keeps its results on the device; noise and synthetic inputs are keyed by the GLOBAL slice index (synth.py), so
the gathered result does not depend on P. The reference has no counterpart (its test path is single process,
batch_size 1: SURVEY.md section 2 'Parallelism strategies'); this replaces what Lightning DDP would do for it.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous block partition; the first n % world ranks get one extra slice"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reconstruct_sharded(n_slices: int, shape_hw: Tuple[int, int], reconstruct_chunk: Callable[[int, int], torch.Tensor],
                        chunk: int = 64, gather: str = "all", group=None) -> Optional[torch.Tensor]:
    """Run `reconstruct_chunk(slice0, count) -> [count,1,H,W]` over this rank's block and gather.

    gather: 'all'  -> every rank returns the full [n_slices,1,H,W] tensor (all_gather),
            'root' -> rank 0 returns it, others None (gather),
            'none' -> each rank returns its own block.
    Uneven blocks are padded to the largest block for the collective and trimmed afterwards.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_slices, rank, world)
    H, W = shape_hw
    parts = []
    for s0 in range(lo, hi, chunk):
        cnt = min(chunk, hi - s0)
        out = reconstruct_chunk(s0, cnt)
        if tuple(out.shape) != (cnt, 1, H, W):
            raise RuntimeError(f"reconstruct_chunk returned {tuple(out.shape)}, expected {(cnt, 1, H, W)}")
        parts.append(out)
    if parts:
        mine = torch.cat(parts, dim=0)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        mine = torch.empty((0, 1, H, W), dtype=torch.float32, device=dev)
    if world == 1 or gather == "none":
        return mine
    sizes = [shard_range(n_slices, r, world) for r in range(world)]
    maxn = max(b - a for a, b in sizes)
    padded = torch.zeros((maxn, 1, H, W), dtype=mine.dtype, device=mine.device)
    padded[: mine.shape[0]] = mine
    if gather == "all":
        full = torch.empty((world * maxn, 1, H, W), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(full, padded, group=group)
        return torch.cat([full[r * maxn: r * maxn + (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)
    if gather == "root":
        bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, bufs, dst=0, group=group)
        if rank != 0:
            return None
        return torch.cat([bufs[r][: (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)
    raise ValueError(f"gather must be 'all', 'root' or 'none', got {gather!r}")


def residual_maps_sharded(engine, n_slices: int, H: int, W: int, *, seed_inputs: int, seed_cond: int, seed_noise: int,
                          t_start: int, chunk: int = 64, gather: str = "all", progress=None) -> Optional[torch.Tensor]:
    """BASELINE config 4: synthetic slices x in (0,1), reconstruct each from noise, gather |x - reco|.
    progress: optional callable(slice0, count) called before every chunk (a long run's sign of life)."""
    from . import synth

    dev = engine.device

    def run(slice0: int, count: int) -> torch.Tensor:
        if progress is not None:
            progress(slice0, count)
        x = torch.from_numpy(synth.synth_slices(seed_inputs, slice0, count, H, W)).to(dev)
        cond = torch.from_numpy(synth.synth_cond(seed_cond, slice0, count)).to(dev)
        x_T = engine.noise_fill(count, H, W, seed=seed_noise, stream_id=synth.STREAM_XT, slice0=slice0)
        reco = engine.reverse(x_T, cond, t_start, seed=seed_noise, slice0=slice0)
        return (x - reco).abs()

    return reconstruct_sharded(n_slices, (H, W), run, chunk=chunk, gather=gather)
