"""Batch sharding across the GPUs of one node.

Environment instances never interact, so the batch is cut into contiguous slices, one engine handle
per device, and there is NO data-path collective (no RCCL): the only cross-rank operations are an
optional host-side gather of results and the benchmark's barrier/clock.  Instance `i` of the global
batch is identical however the batch is sharded: the generator seeds env i with `seed + i` and the
lifelong target streams are keyed by the global env index (`env_index_base`).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

_gloo_group = None


def shard_bounds(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first global env index, number of envs) of `rank`'s contiguous slice; the first
    `global_batch % world_size` ranks hold one extra env."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, rem = divmod(int(global_batch), int(world_size))
    count = q + (1 if rank < rem else 0)
    start = rank * q + min(rank, rem)
    return start, count


def _host_group():
    """A gloo group for host-side gathers (created lazily when the default backend is RCCL)."""
    global _gloo_group
    if dist.get_backend() == "gloo":
        return None
    if _gloo_group is None:
        _gloo_group = dist.new_group(backend="gloo")
    return _gloo_group


def gather_to_host(local: torch.Tensor, global_batch: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Host-side gather of per-rank batch slices (dim 0) into one CPU tensor on rank `dst`."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local.detach().cpu()
    if local.dim() == 0:
        raise ValueError("gather_to_host gathers along the batch axis (dim 0)")
    world, rank = dist.get_world_size(), dist.get_rank()
    host = local.detach().cpu().contiguous()
    as_bool = host.dtype == torch.bool  # gloo has no bool: flags travel as bytes
    if as_bool:
        host = host.view(torch.uint8)
    group = _host_group()
    counts = [shard_bounds(global_batch, world, r)[1] for r in range(world)]
    if host.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {host.shape[0]} envs, its slice of {global_batch} is {counts[rank]}")
    widest = max(counts)  # gloo's gather wants equal shapes: pad ragged slices, trim after
    if host.shape[0] < widest:
        pad = torch.zeros((widest - host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        host = torch.cat([host, pad], dim=0)
    if rank == dst:
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.gather(host, gather_list=parts, dst=dst, group=group)
        whole = torch.cat([part[:c] for part, c in zip(parts, counts)], dim=0)
        return whole.view(torch.bool) if as_bool else whole
    dist.gather(host, gather_list=None, dst=dst, group=group)
    return None


def make_sharded_env(grid_config, global_batch: int, device=None, **kwargs):
    """This rank's VecPogema over its slice of `global_batch` (one process per GPU)."""
    from .vec_env import VecPogema
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    start, count = shard_bounds(global_batch, world, rank)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return VecPogema(grid_config, batch=count, device=device, env_index_base=start, **kwargs)
