"""Where VecPogema's observation buffers come from: the placement policy, kept apart from the environment itself.

On MI355X the position of the observation tensor in HBM decides up to a fifth of the step time, so
the engine allocates it: zone-spread buffers from the engine's pool (`pgx_buffers_*`, buffers.ZoneBuffers) where a walk
is allowed and finds a second zone, a probe-only placement on a shared device, plain torch memory otherwise -- plus the
candidate timing, the XCD share tuning, the recycling output allocator's set-up, the rollout ring and the shelf for the
buffers of closed environments.  `PlacementMixin` is mixed into `VecPogema`; it uses the environment's handle, shapes and
`placement_budget_gib`, and leaves `self.placement` (what happened, for bench lines and tests) and `self._zone_ptrs` behind.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib


def choose_buffers(zone_us, plain_us, n, margin=0.98):
    """Which n of the timed candidates to keep: the n fastest pool ('zone') buffers, each replaced by one of the
    allocator's own ('torch') buffers only where that one is clearly faster (by more than 1 - margin) than the slowest
    pool buffer still chosen -- equal times keep the pool.  Both lists sorted ascending; returns [(kind, index), ...]."""
    chosen = [("zone", i) for i in range(min(n, len(zone_us)))]
    time_of = {"zone": zone_us, "torch": plain_us}
    for j, t in enumerate(plain_us):
        if len(chosen) < n:
            chosen.append(("torch", j))
            continue
        worst = max(chosen, key=lambda c: time_of[c[0]][c[1]])
        if t < margin * time_of[worst[0]][worst[1]]:
            chosen[chosen.index(worst)] = ("torch", j)
    return chosen


class PlacementMixin:
    """Buffer placement of a VecPogema (see the module docstring).  Expects of the class it is mixed into: `_handle`, `_lib`,
    `device`, `device_index`, `obs_shape`, `obs_dtype`, `batch`, `num_agents`, `placement_probe`, `placement_budget_gib`,
    `single_buffer`, `_has_state()`, `_stream()`; maintains `placement`, `_zone_ptrs`, `_recycler`, `_rollout_pools`."""

    def _shelf_key(self):
        return (self.device_index, tuple(self.obs_shape), self.obs_dtype)

    def _park_buffers(self):
        """reuse_buffers='recycle': zone-spread observation buffers that nobody references any more go to the process-wide
        shelf (buffers.ParkedBuffers) for the next environment of this shape instead of being unmapped."""
        rec, pl = getattr(self, "_recycler", None), getattr(self, "placement", None) or {}
        if rec and self._zone_ptrs:
            from .buffers import ParkedBuffers
            try:
                torch.cuda.current_stream(self.device).synchronize()  # nothing of this engine may still be writing
                n = self._recycle_sets(int(np.prod(self.obs_shape)) * _lib.obs_elem_bytes(self.obs_dtype))
                ParkedBuffers.park(self._shelf_key(), rec.retire(), pl, n)  # only zone-pool buffers, only a whole set
            except Exception:
                pass

    # Placement of the double-buffered observation tensors (reuse_buffers=True).  Physical HBM on MI355X falls into a
    # few large zones; a store stream confined to one zone sustains ~5.5 TB/s, the same stream with half of its bytes in
    # another zone ~6.9 TB/s (profiles/r2/placement_*.txt).  A plain allocation is physically
    # compact -- one zone, unless it straddles a boundary by luck (round 1 searched for such lucky buffers by timing up
    # to 64 candidates).  The engine's buffer pool (pgx_buffers_create) REQUESTS the placement instead: each buffer is
    # one virtual range whose second half is backed by another zone, verified by timing.  Buffers below 128 MiB (a
    # repeated stream of that size is absorbed by the Infinity Cache) come from torch's allocator; configs[3]'s two
    # alternating 190 MB buffers exceed the cache together and gain 2-3 % from the pool.
    PLACEMENT_MIN_BYTES = int(os.environ.get("PGX_ZONE_MIN_MB", "128")) << 20

    EXCLUSIVE_FREE_FRACTION = 0.90

    PROBE_ONLY_GIB = 0.01  # a budget below one spacer: pgx_buffers_create times its first probe pair and walks nowhere

    def _walk_policy(self):
        """-> (GiB the zone walk may hold; 0 = no walk, why, explicit?) -- class docstring, `placement_budget_gib`."""
        from .buffers import ZoneBuffers
        b = self.placement_budget_gib
        free, total = torch.cuda.mem_get_info(self.device_index)
        half = min(ZoneBuffers.ALL_GIB, 0.5 * free / float(1 << 30))
        if b == "all":
            return float(os.environ.get("PGX_ZONE_SPACER_GIB", ZoneBuffers.ALL_GIB)), "explicit: all", True
        if b == "half":
            return half, "explicit: half of the free memory", True
        if b is not None:
            return float(b), f"explicit: {float(b):g} GiB", True
        env = os.environ.get("PGX_ZONE_SPACER_GIB")
        if env is not None:
            return float(env), f"PGX_ZONE_SPACER_GIB={env}", True
        if free < self.EXCLUSIVE_FREE_FRACTION * total:
            return self.PROBE_ONLY_GIB, (f"probe only, nothing held: device {self.device_index} is shared or already loaded "
                                         f"({100.0 * free / total:.0f} % of its memory free, < {100 * self.EXCLUSIVE_FREE_FRACTION:.0f} %)"), False
        return half, "auto: the device looks exclusively ours -> half of the free memory", False

    def _pick_obs_buffers(self, n: Optional[int] = None):
        obs_bytes = int(np.prod(self.obs_shape)) * _lib.obs_elem_bytes(self.obs_dtype)
        if n is None:
            n = 1 if self.single_buffer else 2
        if not self.placement_probe or obs_bytes < self.PLACEMENT_MIN_BYTES:
            return self._plain_obs_buffers(n)
        from .buffers import ParkedBuffers, walk_lock
        parked = ParkedBuffers.claim(self._shelf_key(), n)
        if parked is not None:  # buffers a closed environment of this shape left behind: no walk (as _build_recycler)
            bufs, info = parked
            self._zone_ptrs = {t.data_ptr() for t in bufs}
            self.placement = dict(info, method="pgx_buffers (two HBM zones per buffer; taken over from a closed environment)")
            if self.batch >= 2048 and self._has_state():
                self.placement.update(self.tune_xcd_shares(bufs[0], bufs[-1] if n > 1 else None))
            return bufs
        budget, why, explicit = self._walk_policy()
        if budget <= 0.0:
            return self._plain_obs_buffers(n, policy=why)
        if budget < 1.0:
            return self._probe_only_obs_buffers(n, why)
        failed = self._walk_known_to_fail(budget)
        if failed is not None:
            return self._probe_only_obs_buffers(n, failed)
        with walk_lock(self.device_index, wait=explicit) as mine:
            if not mine:  # somebody else is walking this device right now: it is not ours alone
                return self._probe_only_obs_buffers(n, f"probe only, nothing held: another process is walking device {self.device_index}")
            self._budget_now = budget
            bufs = self._pick_obs_buffers_walk(n, obs_bytes)
            if self.placement is not None:
                self.placement.setdefault("policy", why)
            return bufs

    def _walk_known_to_fail(self, budget: float) -> Optional[str]:
        """The policy text of the negative cache (buffers.WalkVerdicts): an earlier engine of this process walked this
        device with at least this budget and found no second zone -> do not hold the memory again, probe only."""
        from .buffers import WalkVerdicts
        v = WalkVerdicts.failed(self.device_index, budget)
        if v is None:
            return None
        return (f"probe only, nothing held: an earlier walk of this process over {v['budget_gib']:.0f} GiB "
                f"({v['candidates']} candidates) found no second zone on device {self.device_index} (negative cache)")

    def _probe_only_obs_buffers(self, n: int, why: str):
        """The co-tenant-safe form of the placement (a shared or loaded device): exactly the n buffers the engine needs, built
        where the allocator stands, after ONE probe pair told whether that spot straddles two zones -- no spacers, no spare
        buffers, no timing tensors, no cache flush, no device-wide synchronisation.  Zone buffers if it does, torch's own
        memory if not."""
        from .buffers import ZoneBuffers
        try:
            pool = ZoneBuffers(self.obs_shape, self.obs_dtype, self.device, count=n, max_spacer_gib=self.PROBE_ONLY_GIB,
                               sync_device=False)
        except _lib.PgxError as e:
            return self._plain_obs_buffers(n, fallback=str(e), policy=why)
        if not pool.info["spread"]:
            probe = {k: pool.info[k] for k in ("same_zone_us", "final_us")}  # what a bare store stream does on this box
            del pool
            return self._plain_obs_buffers(n, policy=why + " -- the allocator does not stand between two zones", probe=probe)
        self.placement = dict(pool.info, method="pgx_buffers (two HBM zones per buffer)", pools_tried=1, policy=why,
                              chosen=["zone"] * n)
        bufs = list(pool.tensors)
        self._zone_ptrs = {t.data_ptr() for t in bufs}
        if self.batch >= 2048 and self._has_state():
            self.placement.update(self.tune_xcd_shares(bufs[0], bufs[-1] if n > 1 else None))
        return bufs

    def _pick_obs_buffers_walk(self, n: int, obs_bytes: int):
        # The walk's verdict does not always carry over to the real buffers (halves may straddle a boundary), and some
        # boxes show no zones at all: ask the pool for two buffers more than needed, time the observation stream itself
        # into each of them and into a few buffers as torch's allocator hands them out, keep the fastest n and give
        # the rest back.
        spare = self.SPARE_BUFFERS if self._has_state() else 0
        from .buffers import WalkVerdicts
        try:
            pools = [self._zone_pool(n + spare)]
        except _lib.PgxError as e:  # e.g. a concurrent allocation took the memory during the walk: plain buffers
            return self._plain_obs_buffers(n, fallback=str(e))
        WalkVerdicts.note_walk(self.device_index, pools[0].info, getattr(self, "_budget_now", 0.0))
        # candidates: (observation pass [us], order, kind, tensor, pool index, index inside the pool)
        cands = [(self._time_observe(t), i, "zone", t, 0, i) for i, t in enumerate(pools[0].tensors)]
        # The probe's promise, scaled to this tensor: if the n-th best buffer misses it by 10 % the fast stretch was
        # narrower than the buffers -- build another pool further on (at most two more) and choose among all of them.
        retry = float(os.environ.get("PGX_POOL_RETRY", "1.10"))
        while spare and len(pools) < 3:
            info = pools[-1].info
            nth = sorted(c[0] for c in cands)[n - 1]
            if not info["spread"] or info["final_us"] <= 0 or nth <= retry * info["final_us"] * obs_bytes / (2 * (384 << 20)):
                break
            try:
                pools.append(self._zone_pool(n + spare, skip_gib=info["spacer_gib"] + 16.0))
            except _lib.PgxError:
                break  # keep what the first pool gave
            k = len(pools) - 1
            cands += [(self._time_observe(t), len(cands) + i, "zone", t, k, i) for i, t in enumerate(pools[k].tensors)]
        self.placement = dict(pools[0].info, method="pgx_buffers (two HBM zones per buffer)", pools_tried=len(pools))
        if self._has_state():
            k = self.PLAIN_CANDIDATES if obs_bytes < (1 << 30) else self.PLAIN_CANDIDATES // 2
            plain = [torch.empty(self.obs_shape, dtype=self.obs_dtype, device=self.device) for _ in range(k)]
            cands += [(self._time_observe(t), len(cands) + i, "torch", t, -1, -1) for i, t in enumerate(plain)]
            del plain
        cands.sort(key=lambda c: (c[0], c[1]))
        zone = [c for c in cands if c[2] == "zone"]
        other = [c for c in cands if c[2] != "zone"]
        picks = choose_buffers([c[0] for c in zone], [c[0] for c in other], n)
        chosen = [(zone if kind == "zone" else other)[i] for kind, i in picks]
        self._zone_ptrs = {c[3].data_ptr() for c in chosen if c[2] == "zone"}
        self.placement.update(chosen=[c[2] for c in chosen], observe_us=[round(c[0], 1) for c in chosen],
                              observe_us_zone=[round(c[0], 1) for c in zone],
                              observe_us_torch_best=round(min((c[0] for c in other), default=0.0), 1))
        if n == 2 and len(other) >= 2 and any(c[2] == "zone" for c in chosen):
            # what counts is the PAIR written in turn (two buffers that together exceed the Infinity Cache behave
            # differently from one that fits it, and single-buffer timings of < 256 MiB tensors all look alike):
            # the chosen pair against two of torch's own buffers
            pair_chosen = self._time_observe(chosen[0][3], chosen[1][3])
            pair_plain = self._time_observe(other[0][3], other[1][3])
            self.placement.update(observe_pair_us=round(pair_chosen, 1), observe_pair_us_torch=round(pair_plain, 1))
            if pair_plain < 0.98 * pair_chosen:
                chosen = other[:2]
                self._zone_ptrs = set()
                self.placement.update(chosen=[c[2] for c in chosen], observe_us=[round(c[0], 1) for c in chosen])
        kept = {(c[4], c[5]) for c in chosen if c[2] == "zone"}
        result = [c[3] for c in chosen]
        self.placement.update(self.tune_xcd_shares(result[0], result[-1] if n > 1 else None))
        del cands, other, zone, chosen
        for k, pool in enumerate(pools):  # a pool none of whose buffers is kept dies with its last reference
            for i in range(n + spare):
                if (k, i) not in kept:
                    pool.drop(i)
        del pools
        torch.cuda.empty_cache()
        return result

    def _zone_pool(self, count: int, skip_gib: float = 0.0):
        from .buffers import ZoneBuffers
        budget = getattr(self, "_budget_now", 0.0)
        return ZoneBuffers(self.obs_shape, self.obs_dtype, self.device, count=count, max_spacer_gib=budget, skip_gib=skip_gib,
                           sync_device=budget >= 1.0)  # (probe-only budgets: no device-wide synchronisation either)

    def _plain_obs_buffers(self, n: int, fallback: Optional[str] = None, policy: Optional[str] = None,
                           probe: Optional[dict] = None):
        """Observation buffers as torch's allocator hands them out (small tensors, probe switched off, no walk on a shared
        device, failed walk)."""
        if policy is None:
            policy = ("the walk failed (see `fallback`): torch's allocator instead" if fallback is not None
                      else "no walk: observation tensor below 128 MiB or placement_probe off")
        self.placement = {"spread": False, "method": "torch allocator", "candidates": 0, "budget_gib": 0.0, "policy": policy}
        self._zone_ptrs = set()
        if probe:
            self.placement.update(probe)
        if fallback is not None:
            self.placement["fallback"] = fallback
            torch.cuda.empty_cache()
        bufs = [torch.empty(self.obs_shape, dtype=self.obs_dtype, device=self.device) for _ in range(n)]
        if self.placement_probe and self.batch >= 2048:
            self.placement.update(self.tune_xcd_shares(bufs[0], bufs[-1] if n > 1 else None))
        return bufs

    @staticmethod
    def _recycle_sets(obs_bytes: int) -> int:
        """Output sets of reuse_buffers='recycle'.  A policy loop holds one observation while the next is written: two
        sets serve it.  Tensors of 64-256 MiB get exactly two (two alternating 190 MB tensors still sit partly in the
        256 MiB Infinity Cache, three do not: configs[3] 36.0 vs 41.0 us per step, profiles/r3/recycle_modes.txt); all
        others a third one for callers that keep (obs, next_obs) pairs."""
        return 2 if (64 << 20) <= obs_bytes < (256 << 20) else 3

    def _build_recycler(self):
        """reuse_buffers='recycle': False (fresh tensors) where the storage-count hook is missing or the walk failed."""
        from .buffers import RecyclingOutputs
        if not RecyclingOutputs.available():
            return False
        obs_bytes = int(np.prod(self.obs_shape)) * _lib.obs_elem_bytes(self.obs_dtype)
        n = self._recycle_sets(obs_bytes)
        if not self.placement_probe or obs_bytes < self.PLACEMENT_MIN_BYTES:
            bufs = self._plain_obs_buffers(n)  # torch's own memory, XCD shares tuned for launches of >= 2048 envs
        else:
            from .buffers import ParkedBuffers
            parked = ParkedBuffers.claim(self._shelf_key(), n)
            if parked is not None:  # buffers a closed environment of this shape left behind: no walk
                bufs, info = parked
                self._zone_ptrs = {t.data_ptr() for t in bufs}
                self.placement = dict(info, method="pgx_buffers (two HBM zones per buffer; taken over from a closed environment)")
                if self.batch >= 2048:
                    self.placement.update(self.tune_xcd_shares(bufs[0], bufs[-1] if n > 1 else None))
            else:
                bufs = self._pick_obs_buffers(n)   # zone walk, candidates timed, the best n kept
        return RecyclingOutputs(bufs, self.batch, self.num_agents, zone_ptrs=self._zone_ptrs)

    def tune_xcd_shares(self, obs: torch.Tensor, obs_alt: Optional[torch.Tensor] = None, rounds: int = 6) -> dict:
        """pgx_xcd_tune: shift work between the 8 XCDs until they finish a launch together (the odd XCDs get through
        their streams 5-15 % slower); keeps the shares with the shortest observation pass, equal shares included."""
        if not self._has_state():
            return {}
        eq, tuned = C.c_float(0.0), C.c_float(0.0)
        _lib.check(self._lib.pgx_xcd_tune(self._handle, obs.data_ptr(), obs_alt.data_ptr() if obs_alt is not None else None,
                                          int(rounds), C.byref(eq), C.byref(tuned), self._stream()))
        shares = (C.c_int32 * 8)()
        _lib.check(self._lib.pgx_xcd_shares(self._handle, shares))
        return {"xcd_shares": list(shares), "observe_us_equal_shares": round(eq.value, 1), "observe_us_tuned_shares": round(tuned.value, 1)}

    PLAIN_CANDIDATES = 8

    SPARE_BUFFERS = 4

    def _time_observe(self, obs: torch.Tensor, obs_alt: Optional[torch.Tensor] = None) -> float:
        """Average duration [us] of the observation stream into `obs` -- or into `obs` and `obs_alt` in turn
        (pgx_time_observe_pair; needs an installed state)."""
        if not self._has_state():
            return 0.0
        us = C.c_float(0.0)
        if obs_alt is not None:
            _lib.check(self._lib.pgx_time_observe_pair(self._handle, obs.data_ptr(), obs_alt.data_ptr(), 8, C.byref(us), self._stream()))
            return float(us.value)
        _lib.check(self._lib.pgx_time_observe(self._handle, obs.data_ptr(), 3, C.byref(us), self._stream()))
        return float(us.value)

    def _ring_from_recycler(self, slots: int, obs_bytes: int):
        """rollout()'s observation ring out of the output sets of reuse_buffers='recycle' (VERDICT r5 next #4): `slots` idle
        sets are taken like any other hand-out and stay out for as long as the returned ring tensor (or a view of it) is
        alive -- step() then serves from the remaining sets or fresh tensors, and nothing is overwritten behind the
        caller's back.  The buffers need not be neighbours: the kernel addresses slot k at base + k * stride, so any two
        buffers form a ring (stride = their distance), more only when they happen to lie at equal distances.
        -> (ring [slots, *obs_shape] strided view, stride in bytes) or None (not in recycle mode, more slots than sets, sets
        in use, unequal distances): the caller falls back to a ring of its own (`_build_rollout_ring`)."""
        if not self.recycle or self._recycler is False:
            return None
        if self._recycler is None:
            if not self._has_state() or torch.cuda.is_current_stream_capturing():
                return None
            self._recycler = self._build_recycler()
        rec = self._recycler
        if not rec or slots > len(rec) or rec.free_sets() < slots:
            return None
        taken = [rec.take() for _ in range(slots)]
        if any(s is None for s in taken):
            return None
        taken.sort(key=lambda s: s[0].data_ptr())
        ptrs = [s[0].data_ptr() for s in taken]
        stride = ptrs[1] - ptrs[0] if slots > 1 else obs_bytes
        if stride < obs_bytes or stride % 16 or any(ptrs[i + 1] - ptrs[i] != stride for i in range(slots - 1)):
            return None
        from .buffers import _Cai, _TYPESTR
        item = _lib.obs_elem_bytes(self.obs_dtype)
        dense = [item]
        for n in reversed(tuple(self.obs_shape)[1:]):
            dense.insert(0, dense[0] * n)
        ring = torch.as_tensor(_Cai(ptrs[0], (slots,) + tuple(self.obs_shape), _TYPESTR[self.obs_dtype], taken,
                                    strides=(stride,) + tuple(dense)), device=self.device)
        if ring.data_ptr() != ptrs[0] or ring.stride(0) * item != stride:
            return None
        return (ring if ring.dtype == self.obs_dtype else ring.view(self.obs_dtype)), stride

    def _build_rollout_ring(self, slots: int, obs_bytes: int):
        """(pool, ring view) of `slots` zone-spread observation slots for rollout(), or None (no walk under the policy of
        `placement_budget_gib`, or a failed one).  With a walk, two buffers more than needed are built; the run of `slots`
        consecutive ones into which the observation stream itself is fastest becomes the ring, the others are given back.
        Probe only (shared or loaded device, a busy walk lock, a walk this process has already seen fail): exactly `slots`
        buffers where the allocator stands, no timing pass, no drops, nothing held -- as `_probe_only_obs_buffers`."""
        from .buffers import WalkVerdicts, walk_lock
        budget, why, explicit = self._walk_policy()
        if budget <= 0.0:
            if self.placement is None:
                self.placement = {"spread": False, "method": "torch allocator", "candidates": 0, "budget_gib": 0.0, "policy": why}
            return None
        if budget >= 1.0:
            failed = self._walk_known_to_fail(budget)
            if failed is not None:
                budget, why = self.PROBE_ONLY_GIB, failed
        if budget < 1.0:
            return self._probe_only_ring(slots, why)
        with walk_lock(self.device_index, wait=explicit) as mine:
            if not mine:
                return self._probe_only_ring(slots, f"probe only, nothing held: another process is walking device {self.device_index}")
            self._budget_now = budget
            retry = float(os.environ.get("PGX_POOL_RETRY", "1.10"))
            pool, times, start, skip = None, None, 0, 0.0
            for attempt in range(3):  # as in _pick_obs_buffers: try further on while the ring misses the probe's promise
                try:
                    cand = self._zone_pool(slots + 2, skip_gib=skip)
                except _lib.PgxError as e:  # failed walk (memory taken meanwhile): keep an earlier pool, if any
                    if pool is None:
                        self.placement = {"spread": False, "method": "torch allocator", "candidates": 0,
                                          "budget_gib": round(budget, 1), "policy": why, "fallback": str(e)}
                    break
                if attempt == 0:
                    WalkVerdicts.note_walk(self.device_index, cand.info, budget)
                ct = [self._time_observe(t) for t in cand.tensors]
                cs = min(range(3), key=lambda s: (max(ct[s:s + slots]), s))
                if pool is None or max(ct[cs:cs + slots]) < max(times[start:start + slots]):
                    pool, times, start = cand, ct, cs
                info = cand.info
                del cand
                if (not info["spread"] or info["final_us"] <= 0 or
                        max(times[start:start + slots]) <= retry * info["final_us"] * obs_bytes / (2 * (384 << 20))):
                    break
                skip = info["spacer_gib"] + 16.0
            if pool is None:
                return None
            ring = pool.ring_view(start, slots)
            for i in range(slots + 2):
                if not start <= i < start + slots:
                    pool.drop(i)
            self.placement = dict(pool.info, method="pgx_buffers (two HBM zones per buffer)", policy=why,
                                  observe_us=[round(t, 1) for t in times[start:start + slots]],
                                  observe_us_zone=[round(t, 1) for t in times])
            if "xcd_shares" not in self.placement and self._bufs is None:
                self.placement.update(self.tune_xcd_shares(ring[0], ring[1] if slots > 1 else None))
            return pool, ring

    def _probe_only_ring(self, slots: int, why: str):
        from .buffers import ZoneBuffers
        plain = {"spread": False, "method": "torch allocator", "candidates": 0, "budget_gib": 0.0, "policy": why}
        try:
            pool = ZoneBuffers(self.obs_shape, self.obs_dtype, self.device, count=slots, max_spacer_gib=self.PROBE_ONLY_GIB,
                               sync_device=False)
        except _lib.PgxError as e:
            self.placement = dict(plain, fallback=str(e))
            return None
        if not pool.info["spread"]:
            self.placement = dict(plain, policy=why + " -- the allocator does not stand between two zones",
                                  same_zone_us=pool.info["same_zone_us"], final_us=pool.info["final_us"])
            del pool
            return None
        self.placement = dict(pool.info, method="pgx_buffers (two HBM zones per buffer)", policy=why)
        return pool, pool.ring_view(0, slots)
