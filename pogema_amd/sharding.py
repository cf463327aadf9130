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


def _aligned(offset: int, itemsize: int) -> int:
    return -(-offset // itemsize) * itemsize


class HostGather:
    """The engineered form of the host-side gather: per-step outputs of batch-sharded engines land in ONE host tensor per
    field on rank `dst`, with no collective and no intermediate host copy (north_star: "host-side gather only").

    All ranks of a node map one shared-memory segment laid out as `slots` x [field][global env row]; every rank registers
    it with the HIP runtime (page-locked, DMA-able) and copies its own rows of every field straight from HBM into place
    with asynchronous D2H copies on a side stream -- the "gather" is N concurrent DMA streams into disjoint rows plus one
    host barrier (gloo) per finished step.  Rank `dst` reads the global tensors in place.  Fields that lie back to back
    in device memory AND in the segment travel as one copy (one rank: the engine's recycled outputs -- rewards |
    terminated | truncated | is_active are one block -- are one 7-byte-per-agent copy per step).

        gather = HostGather({"rewards": ((A,), torch.float32), "terminated": ((A,), torch.bool), ...}, global_batch)
        ticket = gather.start(rewards=rew, terminated=term, ...)   # enqueue: returns at once, the engine's stream is not held up
        ... next step() ...
        host = gather.finish(ticket)     # rank dst: {"rewards": CPU tensor [global_batch, A], ...}; other ranks: None

    A returned tensor is a view of slot `ticket % slots`: it is overwritten by the start() call `slots` steps later
    (default 2: consume step t's outputs before starting step t+2, the contract of `reuse_buffers=True`).
    Without a HIP device (CPU tests, gloo) the same code runs with plain host copies.  `shared=False`: private pinned
    staging per rank + a padded gloo gather (the portable fallback, also taken when the segment cannot be registered).
    Not thread-safe; one instance per consumer."""

    def __init__(self, fields, global_batch: int, dst: int = 0, slots: int = 2, shared: bool = True, device=None):
        self.fields = {k: (tuple(shape), dtype) for k, (shape, dtype) in fields.items()}
        self.global_batch, self.dst, self.slots = int(global_batch), int(dst), int(slots)
        if self.slots < 1 or not self.fields:
            raise ValueError("HostGather needs at least one field and one slot")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.start_row, self.count = shard_bounds(self.global_batch, self.world, self.rank)
        self._group = _host_group() if (dist.is_initialized() and self.world > 1) else None
        self._cuda = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
        self._device = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if self._cuda else torch.device("cpu"))
        # layout of one slot: fields in the order given, each [global_batch, *shape], naturally aligned, no other gaps
        self._layout, off = {}, 0
        for name, (shape, dtype) in self.fields.items():
            item = torch.empty((), dtype=dtype).element_size()
            row = item
            for d in shape:
                row *= int(d)
            off = _aligned(off, item)
            self._layout[name] = (off, row)
            off += row * self.global_batch
        self.slot_bytes = _aligned(off, 64)
        self.nbytes = self.slot_bytes * self.slots
        self._registered = False
        self._mm = None
        self.shared = bool(shared) and self.world > 1
        self.mode = None
        if self.shared:
            try:
                self._host = self._map_shared()
                self.mode = "shared segment"
            except Exception as exc:  # noqa: BLE001  (no /dev/shm, registration refused, ...): every rank must agree
                self._host, self._why_not_shared = None, repr(exc)
            ok = torch.tensor([1 if self._host is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self._group)
            if int(ok.item()) == 0:
                self._unmap()
                self.shared = False
        if not self.shared:
            self._host = self._private_staging()
            self.mode = "private staging" + (" + gloo gather" if self.world > 1 else "")
        self._stream = torch.cuda.Stream(device=self._device) if self._cuda else None
        self._events = [None] * self.slots
        self._inflight = [None] * self.slots  # source tensors of a slot's copies, referenced until finish() / slot reuse:
        #                                       a recycling producer (reuse_buffers='recycle') must not hand them out again
        self._seq = 0
        self._done = -1
        self.copies_per_step = None  # statistics: D2H copies the last start() issued (after merging)

    # -- memory ---------------------------------------------------------------------------------
    def _map_shared(self) -> torch.Tensor:
        """One file in /dev/shm, created by rank dst, mapped by everybody, unlinked as soon as all have it (the memory
        lives as long as it is mapped; nothing is left behind if a rank dies later)."""
        import mmap
        import os
        name = [None]
        if self.rank == self.dst:
            name[0] = f"/dev/shm/pgx_gather_{os.getpid()}_{id(self):x}"
            fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_RDWR | getattr(os, "O_NOFOLLOW", 0), 0o600)
            try:
                os.ftruncate(fd, self.nbytes)
            except OSError:
                os.close(fd)
                os.unlink(name[0])
                name[0] = None
                fd = None
        dist.broadcast_object_list(name, src=self.dst, group=self._group)
        err = None
        try:
            if name[0] is None:
                raise OSError("rank dst could not create the segment")
            if self.rank != self.dst:
                fd = os.open(name[0], os.O_RDWR | getattr(os, "O_NOFOLLOW", 0))
            self._mm = mmap.mmap(fd, self.nbytes)
            os.close(fd)
        except OSError as exc:
            err = exc
        dist.barrier(group=self._group)  # everybody has opened it (or failed): the name can go
        if self.rank == self.dst and name[0] is not None:
            try:
                os.unlink(name[0])
            except OSError:
                pass
        if err is not None:
            raise err
        host = torch.frombuffer(self._mm, dtype=torch.uint8)
        if self._cuda:
            self._register(host)
        return host

    def _register(self, host: torch.Tensor):
        rc = torch.cuda.cudart().cudaHostRegister(host.data_ptr(), host.numel(), 0)
        if int(rc) != 0:
            raise RuntimeError(f"hipHostRegister of the shared segment failed ({rc})")
        self._registered = True
        if not host.is_pinned():
            raise RuntimeError("torch does not see the registered segment as pinned memory")

    def _private_staging(self) -> torch.Tensor:
        return torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=self._cuda)

    def _unmap(self):
        if self._registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(self._host.data_ptr())
            except Exception:  # noqa: BLE001
                pass
            self._registered = False
        self._host = None
        if self._mm is not None:
            try:
                self._mm.close()
            except (BufferError, ValueError):
                pass  # a view is still alive: the mapping dies with it
            self._mm = None

    def close(self):
        if self._stream is not None:
            self._stream.synchronize()
        self._views = None
        self._unmap()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- views ----------------------------------------------------------------------------------
    def _field_view(self, slot: int, name: str, rows=None) -> torch.Tensor:
        """[global_batch (or the given row range), *shape] view of `name` in `slot`."""
        shape, dtype = self.fields[name]
        off, row = self._layout[name]
        lo, n = (0, self.global_batch) if rows is None else rows
        base = slot * self.slot_bytes + off + lo * row
        flat = self._host[base:base + n * row]
        if dtype == torch.bool:
            return flat.view(torch.bool).view((n,) + shape)
        return flat.view(dtype).view((n,) + shape)

    # -- the gather -----------------------------------------------------------------------------
    def start(self, **tensors) -> int:
        """Enqueue the copies of this rank's rows (device tensors [local batch, *shape] for every field, all of them) after
        everything enqueued so far on the current stream; returns the ticket for finish()."""
        if set(tensors) != set(self.fields):
            raise ValueError(f"start() needs exactly the fields {sorted(self.fields)}")
        slot = self._seq % self.slots
        jobs = []  # (host byte offset, byte count, source tensor or (storage-sharing first tensor, byte count))
        for name in self.fields:
            t = tensors[name]
            shape, dtype = self.fields[name]
            ok_dtype = t.dtype == dtype or {t.dtype, dtype} == {torch.bool, torch.uint8}
            if tuple(t.shape) != (self.count,) + shape or not ok_dtype or not t.is_contiguous():
                raise ValueError(f"{name}: expected a contiguous {dtype} tensor of shape {(self.count,) + shape}, got "
                                 f"{t.dtype} {tuple(t.shape)}")
            off, row = self._layout[name]
            jobs.append([slot * self.slot_bytes + off + self.start_row * row, self.count * row, t, t.data_ptr()])
        merged = [jobs[0]]
        for j in jobs[1:]:
            m = merged[-1]
            same_storage = m[2].untyped_storage().data_ptr() == j[2].untyped_storage().data_ptr()
            if same_storage and m[3] + m[1] == j[3] and m[0] + m[1] == j[0]:
                m[1] += j[1]   # back to back on both sides: one copy
            else:
                merged.append(j)
        self.copies_per_step = len(merged)
        if self._cuda:
            ready = torch.cuda.Event()
            ready.record()                       # the producer stream's position now
            self._stream.wait_event(ready)
            prev = self._events[slot]
            with torch.cuda.stream(self._stream):
                for hoff, nbytes, t, _ in merged:
                    self._host[hoff:hoff + nbytes].copy_(self._bytes_of(t, nbytes), non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._stream)
            for _, _, t, _ in merged:
                t.record_stream(self._stream)    # torch's allocator: the side stream still reads it
            if prev is not None:
                prev.synchronize()               # (slot reuse without finish(): its old sources are released below)
            self._events[slot] = done
            self._inflight[slot] = tuple(tensors.values())
        else:
            for hoff, nbytes, t, _ in merged:
                self._host[hoff:hoff + nbytes].copy_(self._bytes_of(t, nbytes))
        self._seq += 1
        return self._seq - 1

    @staticmethod
    def _bytes_of(t: torch.Tensor, nbytes: int) -> torch.Tensor:
        """`nbytes` raw bytes starting at t's first element (may run past t into its neighbours in the same storage)."""
        st = t.untyped_storage()
        return torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, t.data_ptr() - st.data_ptr(), (nbytes,))

    def finish(self, ticket=None):
        """Waits until the copies of `ticket` (default: the latest start()) have landed on every rank; rank dst gets
        {field: CPU tensor [global_batch, *shape]} (views, see the class docstring), the others None."""
        ticket = self._seq - 1 if ticket is None else int(ticket)
        if not (0 <= ticket < self._seq) or ticket < self._seq - self.slots:
            raise ValueError(f"ticket {ticket} is not one of the last {self.slots} start() calls")
        slot = ticket % self.slots
        if self._cuda and self._events[slot] is not None:
            self._events[slot].synchronize()
        self._inflight[slot] = None
        if self.world > 1 and self.shared:
            dist.barrier(group=self._group)  # every rank's rows are in place (DMA complete + host-visible before it entered)
        if self.world > 1 and not self.shared:
            return self._finish_gloo(slot)
        if self.rank != self.dst:
            return None
        return {name: self._field_view(slot, name) for name in self.fields}

    def _finish_gloo(self, slot: int):
        """Fallback without a shared segment: the private staging slot holds this rank's rows in place; one padded gloo
        gather of the rows of every field (packed) per step."""
        counts = [shard_bounds(self.global_batch, self.world, r)[1] for r in range(self.world)]
        widest = max(counts)
        packed = torch.zeros(sum(row for _, row in self._layout.values()) * widest, dtype=torch.uint8)
        pos = 0
        spans = {}
        for name, (off, row) in self._layout.items():
            n = self.count * row
            base = slot * self.slot_bytes + off + self.start_row * row
            packed[pos:pos + n] = self._host[base:base + n]
            spans[name] = (pos, row)
            pos += widest * row
        if self.rank == self.dst:
            parts = [torch.empty_like(packed) for _ in range(self.world)]
            dist.gather(packed, gather_list=parts, dst=self.dst, group=self._group)
            out = {}
            for name, (shape, dtype) in self.fields.items():
                p0, row = spans[name]
                whole = torch.cat([part[p0:p0 + c * row] for part, c in zip(parts, counts)])
                out[name] = (whole.view(torch.bool) if dtype == torch.bool else whole.view(dtype)).view((self.global_batch,) + shape)
            return out
        dist.gather(packed, gather_list=None, dst=self.dst, group=self._group)
        return None


def start_step_gather(gather: "HostGather", step_result, with_obs: bool = False) -> int:
    """`gather.start` for the tuple `VecPogema.step` returned.  rewards / terminated / truncated / is_active (and obs) are
    the caller's tensors; infos['episode_done'] and infos['metrics'] are ENGINE-owned and rewritten by the next step, so
    they are snapshotted on the producer stream first (two tiny device copies) -- the D2H then cannot race the next step."""
    obs, rewards, terminated, truncated, infos = step_result
    t = {"rewards": rewards, "terminated": terminated, "truncated": truncated, "is_active": infos["is_active"]}
    if "episode_done" in gather.fields:
        t["episode_done"] = infos["episode_done"].clone()
        t["metrics"] = infos["metrics"].clone()
    if with_obs:
        t["obs"] = obs
    return gather.start(**t)


def step_output_fields(env, with_obs: bool = False, with_metrics: bool = True) -> dict:
    """HostGather field spec of one VecPogema step: the small per-agent outputs (7 bytes per agent), the per-env episode
    flags / metrics, optionally the observation tensor."""
    A = env.num_agents
    f = {"rewards": ((A,), torch.float32), "terminated": ((A,), torch.bool), "truncated": ((A,), torch.bool),
         "is_active": ((A,), torch.bool)}
    if with_metrics:
        f["episode_done"] = ((), torch.bool)
        f["metrics"] = ((6,), torch.float32)
    if with_obs:
        f["obs"] = (tuple(env.obs_shape[1:]), env.obs_dtype)
    return f


def make_sharded_env(grid_config, global_batch: int, device=None, **kwargs):
    """This rank's VecPogema over its slice of `global_batch` (one process per GPU)."""
    from .vec_env import VecPogema
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    start, count = shard_bounds(global_batch, world, rank)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return VecPogema(grid_config, batch=count, device=device, env_index_base=start, **kwargs)
