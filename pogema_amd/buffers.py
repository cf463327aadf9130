"""Zone-aware observation buffers (pgx_buffers_* of the C-ABI) as torch tensors.

On MI355X a store stream confined to one physical zone of HBM sustains ~5.5 TB/s, the same stream with half of its
bytes in another zone ~6.9 TB/s (profiles/r2/placement_*.txt).  `ZoneBuffers` asks the engine
for buffers whose two halves are backed by different zones (one contiguous virtual range each) and exposes them as
torch tensors through `__cuda_array_interface__` -- torch only wraps the pointer, the memory belongs to the pool and is
released when the last tensor AND the pool object are gone.
"""
from __future__ import annotations

import atexit
import collections
import contextlib
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib

# (numpy's array interface has no bfloat16: such a buffer is wrapped as float16 and re-viewed, same item size)
_TYPESTR = {torch.float32: "<f4", torch.uint8: "|u1", torch.float16: "<f2", torch.bfloat16: "<f2"}


class _Owner:
    """Keeps the native pool alive; destroyed (device synchronised, memory unmapped) when unreferenced."""

    def __init__(self, lib, handle):
        self._lib, self._handle = lib, handle

    def __del__(self):
        try:
            if self._handle is not None and self._handle.value:
                self._lib.pgx_buffers_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class _Cai:
    def __init__(self, ptr, shape, typestr, owner, strides=None):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": strides}
        self._owner = owner  # the tensor references this object, this object references the pool


class ZoneBuffers:
    """`count` buffers of `shape`/`dtype` on `device`, each spread over two HBM zones when reachable.

    max_spacer_gib: memory the search may hold temporarily while it walks the allocator into another zone (released
    before the constructor returns; the engine additionally keeps 10 % of the free memory untouched).  The next zone can
    be 40 ... > 128 GiB away from where the allocator stands (profiles/r2/placement_walk_scan.txt).  While the walk runs
    (~1-2 s) that memory is unavailable to everybody else on the device, so the DEFAULT is bounded: PGX_ZONE_SPACER_GIB
    if set, otherwise half of the memory that is free right now (`default_spacer_gib`); a caller that owns the device
    (bench.py) passes "all" (= up to 272 GiB, i.e. everything the engine's own 10 % reserve leaves); 0 disables the
    search.  PGX_DEBUG=1 traces the walk."""

    ALL_GIB = 272.0

    @staticmethod
    def default_spacer_gib(index: int) -> float:
        env = os.environ.get("PGX_ZONE_SPACER_GIB")
        if env is not None:
            return float(env)
        free, _total = torch.cuda.mem_get_info(index)
        return min(ZoneBuffers.ALL_GIB, 0.5 * free / float(1 << 30))

    def __init__(self, shape, dtype, device, count=2, max_spacer_gib=None, skip_gib=0.0, sync_device=True):
        if dtype not in _TYPESTR:
            raise ValueError(f"unsupported dtype {dtype}")
        dev = torch.device(device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        if max_spacer_gib is None:
            max_spacer_gib = self.default_spacer_gib(index)
        elif max_spacer_gib == "all":
            max_spacer_gib = float(os.environ.get("PGX_ZONE_SPACER_GIB", self.ALL_GIB))
        lib = _lib.load()
        nbytes = int(np.prod(shape)) * _lib.obs_elem_bytes(dtype)
        handle = C.c_void_p()
        if sync_device:
            torch.cuda.synchronize(index)  # the search times kernels (on a private stream): nothing else should be running
        _lib.check(lib.pgx_buffers_create_at(index, nbytes, int(count), float(skip_gib), float(max_spacer_gib), C.byref(handle)))
        self._owner = _Owner(lib, handle)
        info = _lib.PgxBuffersInfo()
        _lib.check(lib.pgx_buffers_get_info(handle, C.byref(info)))
        self.info = {"spread": bool(info.spread), "candidates": int(info.candidates),
                     "same_zone_us": round(float(info.same_zone_us), 2), "final_us": round(float(info.final_us), 2),
                     "spacer_gib": float(info.spacer_gib), "bytes": int(info.bytes), "count": int(info.count),
                     "buffer_gbs": round(float(info.buffer_gbs), 1), "budget_gib": round(float(max_spacer_gib), 1)}
        self.tensors = []
        for i in range(count):
            ptr = lib.pgx_buffers_ptr(handle, i)
            if not ptr:
                raise RuntimeError("pgx_buffers_ptr returned NULL")
            t = torch.as_tensor(_Cai(ptr, shape, _TYPESTR[dtype], self._owner), device=torch.device("cuda", index))
            if t.data_ptr() != ptr:
                raise RuntimeError("torch copied the zone buffer instead of wrapping it")
            if t.dtype != dtype:
                t = t.view(dtype)
            self.tensors.append(t)
        # all buffers lie in one virtual range at a constant stride: the pool as ONE [count, *shape] tensor whose first
        # axis steps by that stride (the observation ring of VecPogema.rollout)
        self.stride_bytes = int(lib.pgx_buffers_stride(handle))
        item = _lib.obs_elem_bytes(dtype)
        dense = [item]
        for n in reversed(tuple(shape)[1:]):
            dense.insert(0, dense[0] * n)
        self._lib, self._handle = lib, handle
        self._view = (tuple(shape), _TYPESTR[dtype], tuple(dense), item, torch.device("cuda", index), dtype)
        self.ring = self.ring_view(0, count)

    def ring_view(self, start: int, count: int) -> torch.Tensor:
        """Buffers start .. start+count-1 as one [count, *shape] tensor (first axis strided by `stride_bytes`)."""
        shape, typestr, dense, item, dev, dtype = self._view
        ptr = self._lib.pgx_buffers_ptr(self._handle, int(start))
        if not ptr or start + count > len(self.tensors):
            raise IndexError("ring_view outside the pool")
        ring = torch.as_tensor(_Cai(ptr, (count,) + shape, typestr, self._owner, strides=(self.stride_bytes,) + dense), device=dev)
        if ring.data_ptr() != ptr or ring.stride(0) * item != self.stride_bytes:
            raise RuntimeError("torch copied the zone buffers instead of wrapping them")
        return ring if ring.dtype == dtype else ring.view(dtype)

    def drop(self, index: int):
        """Release the memory of buffer `index` (pgx_buffers_drop).  The caller must have dropped every reference to
        `tensors[index]`; `ring` is no longer usable as a whole afterwards."""
        self.tensors[index] = None
        self.ring = None
        _lib.check(self._lib.pgx_buffers_drop(self._handle, int(index)))


def device_identity(device_index: int) -> str:
    """A name for the PHYSICAL device behind `device_index` that processes with different HIP_VISIBLE_DEVICES agree on:
    its UUID, else PCI domain:bus:device, else (no HIP device: CPU tests) the index itself."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        uuid = getattr(props, "uuid", None)
        if uuid:
            ident = f"uuid_{uuid}"
        elif hasattr(props, "pci_bus_id"):
            ident = f"pci_{getattr(props, 'pci_domain_id', 0)}_{props.pci_bus_id}_{getattr(props, 'pci_device_id', 0)}"
        else:
            ident = f"index_{device_index}"
    except Exception:  # noqa: BLE001
        ident = f"index_{device_index}"
    return "".join(ch if ch.isalnum() else "_" for ch in ident)


def _open_lock_file(path: str):
    """The per-device lock file as an fd, or None.  Shared temp dirs are hostile ground: never follow a
    symlink planted under the predictable name, make a file we create usable by every tenant (mode 0666 whatever the
    umask), and when another user's file cannot be opened for writing open it read-only -- flock works on that too."""
    flags = getattr(os, "O_NOFOLLOW", 0) | getattr(os, "O_CLOEXEC", 0)
    try:
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR | flags, 0o666)
        try:
            os.fchmod(fd, 0o666)
        except OSError:
            pass
        return fd
    except FileExistsError:
        pass
    except OSError:
        return None
    for mode in (os.O_RDWR, os.O_RDONLY):
        try:
            return os.open(path, mode | flags)
        except OSError:
            continue
    return None


@contextlib.contextmanager
def walk_lock(device_index: int, wait: bool = False):
    """One zone walk per DEVICE at a time, across processes: an exclusive flock on a per-device file in the temp dir
    (keyed by the device's UUID / PCI bus id, so that processes with different HIP_VISIBLE_DEVICES agree).  Yields True
    when this process holds the lock.  wait=False (the default policy): a busy lock means another process is walking the
    device right now, i.e. the device is shared -- yields False at once and the caller skips its walk (workers
    that start together must not each hold half of the free memory).  wait=True (an explicit budget): block until free,
    because two concurrent walks perturb each other's timings.  Never raises.  A lock file that cannot be used at all
    (another tenant's unreadable file, a symlink, a read-only temp dir) means "cannot tell whether the device is ours":
    the default policy then does NOT walk (yields False -> probe only), an explicit budget walks unlocked as asked."""
    import fcntl
    import tempfile
    fd, held = None, True
    try:
        path = os.path.join(tempfile.gettempdir(), f"pgx_zone_walk_{device_identity(device_index)}.lock")
        fd = _open_lock_file(path)
        if fd is None:
            held = bool(wait)
        else:
            try:
                fcntl.flock(fd, fcntl.LOCK_EX | (0 if wait else fcntl.LOCK_NB))
            except BlockingIOError:
                held = False
            except OSError:
                held = bool(wait)
    except Exception:  # noqa: BLE001
        fd, held = None, bool(wait)
    locked = fd is not None and held
    try:
        yield held
    finally:
        if fd is not None:
            try:
                if locked:
                    fcntl.flock(fd, fcntl.LOCK_UN)
            except OSError:
                pass
            finally:
                os.close(fd)


class WalkVerdicts:
    """Negative cache of the zone walk, per (process, PHYSICAL device): once a walk that used its whole budget has found
    no second zone on a device, later engines of this process do not repeat it (a failed walk holds half of
    the free memory for seconds, and bench.py alone builds four engines) -- they go straight to the probe-only placement.
    Only failures are remembered here (successes live on the ParkedBuffers shelf); a later request with a LARGER budget
    than the one that failed may walk again.  Keyed by `device_identity`, so the verdict follows the GPU, not the index.
    PGX_WALK_NEGATIVE_CACHE=0 switches it off; `clear()` (= pogema_amd.release_cached_buffers()) forgets everything."""

    _lock = threading.Lock()
    _failed = {}  # device identity -> {"budget_gib", "candidates", "same_zone_us", "spacer_gib", "walks"}
    walks = 0     # statistics: full walks this process has run (any outcome), for tests and bench lines
    failed_walks = 0  # ... and how many of them found no second zone

    @staticmethod
    def enabled() -> bool:
        return os.environ.get("PGX_WALK_NEGATIVE_CACHE", "1") not in ("", "0")

    TTL_S = float(os.environ.get("PGX_WALK_NEGATIVE_TTL_S", "900"))
    FREE_GROWTH_GIB = 8.0  # one spacer: that much more free memory than at the failed walk means it may reach further now

    @staticmethod
    def _free_gib(device_index: int):
        try:
            return torch.cuda.mem_get_info(device_index)[0] / float(1 << 30) if torch.cuda.is_available() else None
        except Exception:  # noqa: BLE001
            return None

    @classmethod
    def note_walk(cls, device_index: int, info: dict, budget_gib: float):
        """Record the outcome of a walk (`info` = ZoneBuffers.info of its first pool).  A failure is remembered with what
        the walk actually COVERED (the spacers it held at its end -- a walk cut short by a co-tenant or by low free memory
        says nothing about the stretch it never reached), the free memory at that moment and the time.  Walks
        that end on an allocation error never get here (the pool's constructor raises)."""
        import time
        with cls._lock:
            cls.walks += 1
            if info.get("spread"):
                cls._failed.pop(device_identity(device_index), None)
                return
            cls.failed_walks += 1
            key = device_identity(device_index)
            prev = cls._failed.get(key)
            covered = info.get("spacer_gib")
            covered = float(budget_gib) if covered is None else min(float(budget_gib), float(covered) + 8.0)
            cls._failed[key] = {"budget_gib": max(covered, prev["budget_gib"] if prev else 0.0), "requested_gib": float(budget_gib),
                                "candidates": int(info.get("candidates", 0)), "same_zone_us": info.get("same_zone_us"),
                                "spacer_gib": info.get("spacer_gib"), "walks": (prev["walks"] if prev else 0) + 1,
                                "free_gib": cls._free_gib(device_index), "when": time.monotonic()}

    @classmethod
    def failed(cls, device_index: int, budget_gib: float = 0.0):
        """The remembered failure that makes a walk with `budget_gib` pointless on this device, or None -- also None (and
        forgotten) once the verdict is older than TTL_S or the device has materially more free memory than it had then."""
        if not cls.enabled():
            return None
        import time
        with cls._lock:
            key = device_identity(device_index)
            v = cls._failed.get(key)
            if v is None:
                return None
            free_now = cls._free_gib(device_index)
            stale = time.monotonic() - v.get("when", 0.0) > cls.TTL_S
            grown = (free_now is not None and v.get("free_gib") is not None and free_now > v["free_gib"] + cls.FREE_GROWTH_GIB)
            if stale or grown:
                del cls._failed[key]
                return None
            # (a budget within one 8 GiB spacer of what the failed walk covered reaches nothing it did not)
            return dict(v) if float(budget_gib) <= v["budget_gib"] + 8.0 else None

    @classmethod
    def clear(cls):
        with cls._lock:
            cls._failed.clear()


class ParkedBuffers:
    """Process-wide shelf for the observation buffers of CLOSED environments (reuse_buffers='recycle').

    Picking zone-spread buffers costs a walk of 1-3 s, and a pool's address ranges are never handed back to the driver
    (DESIGN.md section 6), so a process that creates environments in a loop -- sweeps, test suites, evaluation workers --
    would pay the walk and leak address space every time.  Instead `VecPogema.close()` parks the zone-spread buffers
    nobody references any more, still mapped, and the next environment with the same observation tensor on the same
    device takes them over without a walk.

    Only WHOLE sets are kept: a set is exactly the number of buffers an environment of that shape claims, so
    whatever sits on the shelf is usable; a close() that can return fewer (the caller still holds a tensor) parks
    nothing, a set larger than the limit is not parked at all, and eviction drops whole sets, oldest first.  The limit
    is PGX_POOL_CACHE_MB (default 2560: one set of configs[2]; 0 = off) -- memory a caller does NOT get back by closing
    an environment until `clear()` (= `pogema_amd.release_cached_buffers()`, `VecPogema.close(release=True)`) or exit."""

    _lock = threading.Lock()
    _shelf = collections.OrderedDict()  # key -> [(tensors of one set, placement info), ...]

    @staticmethod
    def limit_bytes() -> int:
        return int(os.environ.get("PGX_POOL_CACHE_MB", "2560")) << 20

    @staticmethod
    def _set_bytes(tensors) -> int:
        return sum(t.numel() * t.element_size() for t in tensors)

    @classmethod
    def bytes_parked(cls) -> int:
        with cls._lock:
            return sum(cls._set_bytes(ts) for sets in cls._shelf.values() for ts, _ in sets)

    @classmethod
    def park(cls, key, tensors, info, set_size: int) -> bool:
        """Shelve `tensors` as one set of exactly `set_size` buffers (surplus buffers are simply released); False -- and
        nothing kept -- when fewer were returned, the shelf is off, or the set alone exceeds the limit."""
        limit = cls.limit_bytes()
        tensors = list(tensors)[:set_size]
        if limit <= 0 or set_size < 1 or len(tensors) < set_size or cls._set_bytes(tensors) > limit:
            return False
        with cls._lock:
            cls._shelf.setdefault(key, []).append((tensors, dict(info)))
            cls._shelf.move_to_end(key)
            total = sum(cls._set_bytes(ts) for sets in cls._shelf.values() for ts, _ in sets)
            while total > limit:  # whole sets, oldest key first, oldest set of that key first (never the one just parked)
                old = next(iter(cls._shelf))
                gone, _ = cls._shelf[old].pop(0)
                total -= cls._set_bytes(gone)
                if not cls._shelf[old]:
                    del cls._shelf[old]
            return True

    @classmethod
    def claim(cls, key, n):
        """A parked set of n buffers for `key` as ([tensors], its placement info), or None."""
        with cls._lock:
            sets = cls._shelf.get(key)
            if not sets:
                return None
            for i in range(len(sets) - 1, -1, -1):
                if len(sets[i][0]) == n:
                    tensors, info = sets.pop(i)
                    if not sets:
                        del cls._shelf[key]
                    return tensors, info
            return None

    @classmethod
    def clear(cls):
        with cls._lock:
            cls._shelf.clear()


atexit.register(ParkedBuffers.clear)  # release parked pools while the HIP runtime is still up, not during module teardown


def storage_count_hook():
    """`torch._C._storage_Use_Count` if this torch build has it AND it still means what RecyclingOutputs relies on -- the
    number of live tensors (views included) sharing a storage: +1 per extra tensor or view, back down when they die.  The
    hook is private API, so its meaning is re-checked on a tiny CPU tensor (once per process) instead of trusted by
    version; None -- with one loud warning -- when it is missing or behaves differently, and reuse_buffers='recycle' then
    degrades to fresh tensors per step (correct, slower).  tests/test_buffer_choice.py fails when that happens, so a torch
    upgrade that changes the hook is noticed on the CPU."""
    global _HOOK
    if _HOOK is not _UNSET:
        return _HOOK
    hook, why = getattr(torch._C, "_storage_Use_Count", None), None
    if hook is None:
        why = "torch._C._storage_Use_Count is missing"
    else:
        try:
            master = torch.empty(8, dtype=torch.uint8)
            cdata = master.untyped_storage()._cdata
            base = hook(cdata)
            alias = master.detach()
            one = hook(cdata)
            view = alias[2:4].view(torch.int16)
            two = hook(cdata)
            del alias
            still = hook(cdata)   # the view keeps the storage referenced
            del view
            back = hook(cdata)
            if not (one == base + 1 and two == base + 2 and still == base + 1 and back == base):
                why = f"torch._C._storage_Use_Count no longer counts tensors per storage ({base}, {one}, {two}, {still}, {back})"
        except Exception as exc:  # noqa: BLE001
            why = f"torch._C._storage_Use_Count self-test raised {exc!r}"
    if why is not None:
        import warnings
        warnings.warn(f"pogema_amd: {why} in torch {torch.__version__}; reuse_buffers='recycle' falls back to allocating "
                      f"fresh output tensors per step", RuntimeWarning, stacklevel=2)
        hook = None
    _HOOK = hook
    return hook


_UNSET = object()
_HOOK = _UNSET


class RecyclingOutputs:
    """Placement-aware output allocator: a few complete output SETS handed out as ORDINARY tensors.

    A set is one observation tensor (a zone-spread pool buffer for large tensors, torch's own memory for small ones)
    plus one small block carved into rewards f32 / terminated / truncated / is_active bool [batch, agents].  `take()`
    returns the tensors of a set that nobody references any more, or None while every set is still referenced by the
    caller (who then gets fresh torch tensors).  Nothing is ever overwritten behind the caller's back: a set is handed
    out again only when the last tensor or view on ANY of its members has been dropped -- the semantics of torch's own
    caching allocator, including its caveat: memory is re-used in the order of the stream the engine writes on, so a
    consumer on ANOTHER stream must synchronise before dropping its reference (torch: Tensor.record_stream).

    "Nobody references it" is read off the storages' reference counts (torch._C._storage_Use_Count, 0.15 us each); a
    handed-out tensor is `master.detach()` (1 us) -- together half the host time of five torch.empty calls.  The hook is
    private torch API: `storage_count_hook()` verifies its meaning before it is used, `available()` says whether it
    passed; without it the caller allocates fresh tensors (and has been warned)."""

    @staticmethod
    def available() -> bool:
        return storage_count_hook() is not None

    def __init__(self, obs_tensors, batch: int, agents: int, zone_ptrs=()):
        if not self.available():
            raise RuntimeError("torch._C._storage_Use_Count is missing or changed its meaning in this torch build")
        self._count = storage_count_hook()
        self._zone_ptrs = frozenset(int(p) for p in zone_ptrs)  # observation buffers that come from a zone pool
        dev = obs_tensors[0].device
        n = batch * agents
        self._sets = []
        for obs in obs_tensors:
            block = torch.empty(n * 7 + 16, dtype=torch.uint8, device=dev)  # f32 rewards first (alignment), then 3 x u8
            members = (obs, block[:4 * n].view(torch.float32).view(batch, agents),
                       block[4 * n:5 * n].view(torch.bool).view(batch, agents),
                       block[5 * n:6 * n].view(torch.bool).view(batch, agents),
                       block[6 * n:7 * n].view(torch.bool).view(batch, agents))
            storages = (obs.untyped_storage(), block.untyped_storage())
            self._sets.append((members, storages, tuple(st._cdata for st in storages)))
        del obs, block, members, storages
        self._idle = [tuple(self._count(c) for c in cdata) for _, _, cdata in self._sets]
        self._next = 0
        self.taken = 0      # statistics: sets handed out / requests that found every set in use
        self.misses = 0

    def __len__(self):
        return len(self._sets)

    def obs_pointers(self):
        return [members[0].data_ptr() for members, _, _ in self._sets]

    def _is_idle(self, i: int) -> bool:
        cdata, idle = self._sets[i][2], self._idle[i]
        return self._count(cdata[0]) == idle[0] and self._count(cdata[1]) == idle[1]

    def free_sets(self) -> int:
        return sum(self._is_idle(i) for i in range(len(self._sets)))

    def retire(self):
        """The ZONE-POOL observation buffers nobody outside the pool references any more, removed from circulation (for
        ParkedBuffers when the environment closes); sets that are still referenced stay and die with their last user, and
        buffers from torch's own allocator simply go back to it."""
        idle = [i for i in range(len(self._sets)) if self._is_idle(i)]
        out = [self._sets[i][0][0] for i in idle if self._sets[i][0][0].data_ptr() in self._zone_ptrs]
        for i in reversed(idle):
            del self._sets[i]
            del self._idle[i]
        self._next = 0
        return out

    def take(self, with_obs: bool = True):
        """(obs, rewards, terminated, truncated, is_active) of an unreferenced set -- least recently handed out first --
        or None.  with_obs=False: obs is None (the set's observation buffer stays idle)."""
        n = len(self._sets)
        for k in range(n):
            i = (self._next + k) % n
            if self._is_idle(i):
                self._next = (i + 1) % n
                self.taken += 1
                m = self._sets[i][0]
                return (m[0].detach() if with_obs else None, m[1].detach(), m[2].detach(), m[3].detach(), m[4].detach())
        self.misses += 1
        return None
