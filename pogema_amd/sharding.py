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
    with asynchronous D2H copies on a side stream -- the "gather" is N concurrent DMA streams into disjoint rows.  Ranks
    synchronise through two counters per rank in the segment's header (which ticket has LANDED, which one rank dst has
    RELEASED): no gloo call on the per-step path.  Rank `dst` reads the global tensors in place.  Fields that lie back to
    back in device memory AND in the segment travel as one copy (one rank: the engine's recycled outputs -- rewards |
    terminated | truncated | is_active are one block -- are one 7-byte-per-agent copy per step).

        gather = HostGather({"rewards": ((A,), torch.float32), "terminated": ((A,), torch.bool), ...}, global_batch)
        ticket = gather.start(rewards=rew, terminated=term, ...)   # returns at once, the engine's stream is not held up
        ... next step() ...
        host = gather.finish(ticket)     # rank dst: {"rewards": CPU tensor [global_batch, A], ...}; other ranks: None

    The dependency "copy after the step that produced the data" is kept on the HOST (measured on MI355X / ROCm 7.2,
    profiles/r5/host_gather_variants.txt: a cross-stream hipStreamWaitEvent per step more than doubles the loop time --
    257 vs 113 us per configs[2] step --, an event the host polls costs 5-8 us): start() only records an event behind
    the producer and keeps the source tensors referenced; the copies are issued, with no GPU-side wait, by the next
    start() / finish() call that finds that event complete.  So the copy of step t runs under step t+1's kernel as long
    as the caller keeps one step in the queue -- `start(t); finish(t-1)` or deeper.

    Lifetime of what finish(t) returns: views of slot `t % slots`, valid until rank dst's next finish() call; a slot is
    overwritten only by a ticket `slots` later, and start() of that ticket waits (other ranks) or raises (rank dst itself:
    a caller bug) until dst has released it.  The unpipelined loop (start(t); finish(t)) needs 2 slots, `start(t);
    finish(t-1)` 3 (the default), one more per extra step in flight.
    Without a HIP device (CPU tests) the same code runs with plain host copies.  `shared=False`: private pinned staging
    per rank + a padded gloo gather per step (the portable fallback, also taken when the segment cannot be created or
    registered -- decided collectively).  Not thread-safe; one instance per consumer."""

    HEADER = 64  # bytes per rank in the segment's header: int64 landed, int64 released (rank dst's line only)

    def __init__(self, fields, global_batch: int, dst: int = 0, slots: int = 3, shared: bool = True, device=None,
                 timeout_s: float = 120.0):
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
        self.timeout_s = float(timeout_s)
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
        self._data0 = self.HEADER * self.world
        self.nbytes = self._data0 + self.slot_bytes * self.slots
        self._registered = False
        self._mm = None
        self._ctr = None
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
        self._pending = []                    # tickets whose copies are not issued yet: [ticket, slot, jobs, ready event, sources]
        self._issued = {}                     # ticket -> (done event or None, sources): copies in flight
        self._plans = {}                      # (slot, source addresses) -> copy plan, see start()
        self._views = {}                      # slot -> {field: view}, built once
        self._seq = 0
        self._done = -1                       # latest ticket this rank has finished
        self.copies_per_step = None  # statistics: D2H copies per start() (after merging)

    # -- memory ---------------------------------------------------------------------------------
    def _map_shared(self) -> torch.Tensor:
        """One file in /dev/shm, created by rank dst, mapped by everybody, unlinked as soon as all have it (the memory
        lives as long as it is mapped; nothing is left behind if a rank dies later)."""
        import mmap
        import os
        import numpy as np
        name = [None]
        fd = None
        if self.rank == self.dst:
            name[0] = f"/dev/shm/pgx_gather_{os.getpid()}_{id(self):x}"
            try:
                fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_RDWR | getattr(os, "O_NOFOLLOW", 0), 0o600)
                os.ftruncate(fd, self.nbytes)
            except OSError:
                if fd is not None:
                    os.close(fd)
                    os.unlink(name[0])
                name[0], fd = None, None
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
        # Every rank runs the SAME sequence of collectives whatever fails locally (a rank that raised between the
        # two barriers went on to the all_reduce of __init__ while the healthy ranks sat in the second barrier -- gloo does not
        # match the two, everybody hung until the group timeout): errors are kept, both barriers are always executed, and
        # the error is raised only behind the last one; the all_reduce(MIN) in __init__ then turns it into the common fallback.
        host = None
        if err is None:
            try:
                self._ctr = np.frombuffer(self._mm, dtype=np.int64, count=self._data0 // 8)
                self._ctr[self.rank * 8] = -1          # landed
                if self.rank == self.dst:
                    self._ctr[self.rank * 8 + 1] = -1  # released
                host = torch.frombuffer(self._mm, dtype=torch.uint8)
                if self._cuda:
                    self._register(host)
            except Exception as exc:  # noqa: BLE001  (hipHostRegister / memlock refusal on this rank only, ...)
                err = exc
        dist.barrier(group=self._group)  # every counter is initialised before anybody publishes a ticket
        if err is not None:
            if self._registered and host is not None:
                try:
                    torch.cuda.cudart().cudaHostUnregister(host.data_ptr())
                except Exception:  # noqa: BLE001
                    pass
                self._registered = False
            host = None
            self._ctr = None
            raise err
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
        self._ctr = None
        if self._mm is not None:
            try:
                self._mm.close()
            except (BufferError, ValueError):
                pass  # a view is still alive: the mapping dies with it
            self._mm = None

    def close(self):
        if self._stream is not None:
            self._stream.synchronize()
        self._pending, self._issued, self._views = [], {}, {}
        self._snaps, self._snap_src = {}, {}
        self._unmap()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- views ----------------------------------------------------------------------------------
    def _field_view(self, slot: int, name: str) -> torch.Tensor:
        """[global_batch, *shape] view of `name` in `slot`."""
        shape, dtype = self.fields[name]
        off, row = self._layout[name]
        base = self._data0 + slot * self.slot_bytes + off
        flat = self._host[base:base + self.global_batch * row]
        if dtype == torch.bool:
            return flat.view(torch.bool).view((self.global_batch,) + shape)
        return flat.view(dtype).view((self.global_batch,) + shape)

    # -- cross-rank counters ----------------------------------------------------------------------
    def _spin(self, cond, what: str):
        import os
        import time
        t0, polls = time.monotonic(), 0
        while not cond():
            polls += 1
            if polls % 64 == 0:
                os.sched_yield()
                if polls > 20000:       # a long wait (a slow peer, not the normal microseconds): stop burning the core
                    time.sleep(50e-6)
                if time.monotonic() - t0 > self.timeout_s:
                    raise RuntimeError(f"HostGather rank {self.rank}: waited {self.timeout_s:.0f} s for {what}")

    def _released(self) -> int:
        """Highest ticket rank dst has given up (its views are dead)."""
        if self._ctr is not None:
            return int(self._ctr[self.dst * 8 + 1])
        return self._done - 1

    # -- the gather -----------------------------------------------------------------------------
    def start(self, **tensors) -> int:
        """Registers the copies of this rank's rows (device tensors [local batch, *shape] for every field, all of them)
        behind everything enqueued so far on the current stream; returns the ticket for finish()."""
        if set(tensors) != set(self.fields):
            raise ValueError(f"start() needs exactly the fields {sorted(self.fields)}")
        q = self._seq
        slot = q % self.slots
        if q >= self.slots:  # the slot still holds ticket q - slots: rank dst must have released it
            old = q - self.slots
            self._retire(old)
            if self.rank == self.dst or self._ctr is None:
                if old > self._done - 1:
                    raise ValueError(f"HostGather: start() number {q} would overwrite ticket {old}, whose views are still "
                                     f"valid (latest finished ticket: {self._done}); call finish() first or use more slots "
                                     f"({self.slots} now)")
            else:
                self._spin(lambda: self._released() >= old, f"rank {self.dst} to release ticket {old}")
        # The copy plan (which fields merge into which host range) depends only on the slot and on where the sources lie;
        # a recycling producer hands out the same few sets in turn, so the plan is cached by (slot, source addresses) and
        # the per-field validation runs on a miss only.  (The plan holds no tensors: a cached alias would keep a recycled
        # output set referenced for ever.)
        names = tuple(self.fields)
        # (shape, dtype and contiguity are part of the key -- ADVICE r5: another tensor that re-uses the same addresses with
        # another shape or dtype must not be copied with the old plan's byte counts)
        key = (slot,) + tuple((t.data_ptr(), t.dtype, t.shape, t.is_contiguous()) for t in (tensors[n] for n in names))
        plan = self._plans.get(key)
        if plan is None:
            jobs = []  # [host byte offset, byte count, index of the first source field, device address]
            for i, name in enumerate(names):
                t = tensors[name]
                shape, dtype = self.fields[name]
                ok_dtype = t.dtype == dtype or {t.dtype, dtype} == {torch.bool, torch.uint8}
                if tuple(t.shape) != (self.count,) + shape or not ok_dtype or not t.is_contiguous():
                    raise ValueError(f"{name}: expected a contiguous {dtype} tensor of shape {(self.count,) + shape}, got "
                                     f"{t.dtype} {tuple(t.shape)}")
                off, row = self._layout[name]
                jobs.append([self._data0 + slot * self.slot_bytes + off + self.start_row * row, self.count * row, i, t.data_ptr()])
            merged = [jobs[0]]
            for j in jobs[1:]:
                m = merged[-1]
                same_storage = (tensors[names[m[2]]].untyped_storage().data_ptr() == tensors[names[j[2]]].untyped_storage().data_ptr())
                if same_storage and m[3] + m[1] == j[3] and m[0] + m[1] == j[0]:
                    m[1] += j[1]   # back to back on both sides: one copy
                else:
                    merged.append(j)
            plan = tuple((hoff, nbytes, i) for hoff, nbytes, i, _ in merged)
            if len(self._plans) > 64:
                self._plans.clear()
            self._plans[key] = plan
        self.copies_per_step = len(plan)
        copies = [(hoff, nbytes, self._bytes_of(tensors[names[i]], nbytes)) for hoff, nbytes, i in plan]
        ready = None
        if self._cuda:
            ready = torch.cuda.Event()
            ready.record()  # the producer stream's position now
        # (the sources stay referenced until their copies have landed: a recycling producer must not hand them out again)
        self._pending.append((q, copies, ready, tuple(tensors.values())))
        self._seq += 1
        self._pump()
        return q

    def _retire(self, old: int):
        """This rank's OWN copies of tickets <= `old` have landed (normally long since: a no-op) -- required before the
        slot and the snapshot buffer of ticket `old` are rewritten, also on a rank that never called finish() for it."""
        if self._pending and self._pending[0][0] <= old:
            self._pump(upto=old)
        for k in [k for k in self._issued if k <= old]:
            done, _sources = self._issued.pop(k)
            if done is not None:
                done.synchronize()

    def _pump(self, upto: int = -1):
        """Issues the copies of every pending ticket whose producer has finished (tickets <= `upto`: waits for it)."""
        while self._pending:
            q, copies, ready, sources = self._pending[0]
            if ready is not None and not ready.query():
                if q > upto:
                    return
                ready.synchronize()
            self._pending.pop(0)
            done = None
            if self._cuda:
                with torch.cuda.stream(self._stream):  # no GPU-side dependency: the host has seen the producer finish
                    for hoff, nbytes, src in copies:
                        self._host[hoff:hoff + nbytes].copy_(src, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self._stream)
            else:
                for hoff, nbytes, src in copies:
                    self._host[hoff:hoff + nbytes].copy_(src)
            self._issued[q] = (done, sources)

    def snapshot_pair(self, flags: torch.Tensor, values: torch.Tensor):
        """Copies of a bool/uint8 tensor and a float32 tensor (the engine-owned episode_done [B] and metrics [B, 6]) in ONE
        device launch, back to back in a buffer that belongs to the slot the next start() will use -- safe to rewrite
        because that slot's previous ticket has been finished by then (start() checks).  Views are built once per slot
        and per source pair: the per-step cost is one dictionary look-up and the fused copy."""
        slot = self._seq % self.slots
        if self._seq >= self.slots:
            self._retire(self._seq - self.slots)  # the D2H copy that read this slot's snapshot buffer last time is done
        if not hasattr(self, "_snaps"):
            self._snaps, self._snap_src = {}, {}
        skey = (flags.data_ptr(), values.data_ptr(), flags.numel(), values.numel())
        src = self._snap_src.get(skey)
        if src is None:
            if len(self._snap_src) > 8:
                self._snap_src.clear()
            src = self._snap_src[skey] = [flags.reshape(-1).view(torch.uint8), values.reshape(-1).view(torch.uint8)]
        nf, nv = src[0].numel(), src[1].numel()
        entry = self._snaps.get(slot)
        if entry is None or entry[0] != (nf, nv, flags.device, flags.dtype, values.dtype, tuple(flags.shape), tuple(values.shape)):
            off = _aligned(nf, values.element_size())
            buf = torch.empty(off + nv, dtype=torch.uint8, device=flags.device)
            entry = self._snaps[slot] = ((nf, nv, flags.device, flags.dtype, values.dtype, tuple(flags.shape), tuple(values.shape)),
                                         [buf[:nf], buf[off:off + nv]],
                                         buf[:nf].view(flags.dtype).view(flags.shape),
                                         buf[off:off + nv].view(values.dtype).view(values.shape))
        torch._foreach_copy_(entry[1], src)
        return entry[2], entry[3]

    @staticmethod
    def _bytes_of(t: torch.Tensor, nbytes: int) -> torch.Tensor:
        """`nbytes` raw bytes starting at t's first element (may run past t into its neighbours in the same storage)."""
        st = t.untyped_storage()
        return torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, t.data_ptr() - st.data_ptr(), (nbytes,))

    def finish(self, ticket=None):
        """Waits until the copies of `ticket` (default: the latest start()) have landed on every rank; rank dst gets
        {field: CPU tensor [global_batch, *shape]} (views, see the class docstring), the others None.  Tickets must be
        finished in increasing order (skipping is allowed: finishing t gives up everything before it)."""
        ticket = self._seq - 1 if ticket is None else int(ticket)
        if not (0 <= ticket < self._seq) or ticket < self._seq - self.slots or ticket < self._done:
            raise ValueError(f"ticket {ticket} is not one of the last {self.slots} start() calls (or older than the latest "
                             f"finished one, {self._done})")
        slot = ticket % self.slots
        self._pump(upto=ticket)
        for q in sorted(k for k in self._issued if k <= ticket):
            done, _sources = self._issued.pop(q)
            if done is not None:
                done.synchronize()
        self._done = max(self._done, ticket)
        if self._ctr is not None:
            self._ctr[self.rank * 8] = ticket              # this rank's rows of every ticket <= `ticket` are in place
            if self.rank == self.dst:
                self._ctr[self.rank * 8 + 1] = ticket - 1  # ... and the views of everything before it are given up
                self._spin(lambda: all(int(self._ctr[r * 8]) >= ticket for r in range(self.world)),
                           f"the rows of ticket {ticket} from every rank")
        if self.world > 1 and not self.shared:
            return self._finish_gloo(slot)
        if self.rank != self.dst:
            return None
        views = self._views.get(slot)
        if views is None:
            views = self._views[slot] = {name: self._field_view(slot, name) for name in self.fields}
        return dict(views)

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
            base = self._data0 + slot * self.slot_bytes + off + self.start_row * row
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
    they are snapshotted on the producer stream first -- ONE fused device copy into a per-slot buffer in which they lie
    back to back (and so travel as one D2H copy) -- the D2H then cannot race the next step."""
    obs, rewards, terminated, truncated, infos = step_result
    t = {"rewards": rewards, "terminated": terminated, "truncated": truncated, "is_active": infos["is_active"]}
    if "episode_done" in gather.fields:
        t["episode_done"], t["metrics"] = gather.snapshot_pair(infos["episode_done"], infos["metrics"])
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
