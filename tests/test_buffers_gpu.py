"""GPU: the zone-aware buffer pool (pgx_buffers_*, pogema_amd/buffers.py) -- one contiguous virtual range per buffer,
usable like any device memory, and (when another HBM zone is reachable) verified faster for a store stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def empty_shelf():
    """These tests look at HOW buffers were obtained: no hand-me-downs from environments closed by earlier tests."""
    from pogema_amd.buffers import ParkedBuffers
    ParkedBuffers.clear()
    yield
    ParkedBuffers.clear()


def test_pool_buffers_are_ordinary_device_memory():
    import torch
    from pogema_amd.buffers import ZoneBuffers
    shape = (3072, 64, 3, 11, 11)  # 285 MB: large enough for the zone walk
    pool = ZoneBuffers(shape, torch.float32, "cuda:0", count=2)
    a, b = pool.tensors
    assert a.shape == shape and a.dtype == torch.float32 and a.is_contiguous() and a.data_ptr() != b.data_ptr()
    assert a.data_ptr() % (2 << 20) == 0
    ref = torch.arange(a.numel(), dtype=torch.float32, device="cuda").view(shape) % 977
    a.copy_(ref)            # torch kernels write across the seam between the two physical halves
    b.fill_(3.0)
    assert torch.equal(a, ref) and float(b.sum()) == 3.0 * b.numel()
    assert torch.equal(a.cpu(), ref.cpu())  # D2H copies too
    info = pool.info
    assert info["count"] == 2 and info["bytes"] == a.numel() * 4 and info["same_zone_us"] > 0
    if info["spread"]:
        # accepted because faster than a same-zone pair (the walk's weak criterion: 6 %), or because at the two-zone rate
        # (>= 6.6 TB/s on 768 MiB)
        assert info["final_us"] < 0.94 * info["same_zone_us"] or info["final_us"] <= 122.1
    # the memory outlives the pool object for as long as a tensor references it
    del pool, b
    a.add_(1.0)
    torch.cuda.synchronize()
    assert torch.equal(a, ref + 1.0)


def test_pool_without_search_and_small_sizes():
    import torch
    from pogema_amd.buffers import ZoneBuffers
    pool = ZoneBuffers((3, 5, 7), torch.uint8, "cuda:0", count=3, max_spacer_gib=0)  # < one granule: a single handle
    assert not pool.info["spread"] and pool.info["candidates"] == 0 and len(pool.tensors) == 3
    for k, t in enumerate(pool.tensors):
        t.fill_(k + 1)
    assert [int(t.sum()) for t in pool.tensors] == [105 * (k + 1) for k in range(3)]


def test_engine_parity_on_zone_buffers():
    """VecPogema(reuse_buffers=True) writes its observations into pool buffers: same results as fresh tensors."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=3, collision_system="soft")
    B = 1024  # 1024 x 32 x 1452 B = 47.6 MB per buffer: below the pool threshold -> force it
    envs = []
    for reuse in (False, True):
        env = VecPogema(gc, batch=B, auto_reset=True, reuse_buffers=reuse, placement_budget_gib=8.0)  # explicit: walk
        env.PLACEMENT_MIN_BYTES = 1 << 20
        env.reset(seed=3)
        envs.append(env)
    acts = torch.randint(0, 5, (6, B, 32), device="cuda", dtype=torch.int8)
    for t in range(6):
        o0, r0, te0, tr0, _ = envs[0].step(acts[t])
        o1, r1, te1, tr1, _ = envs[1].step(acts[t])
        assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(te0, te1) and torch.equal(tr0, tr1)
    assert envs[1].placement["method"].startswith("pgx_buffers")
    for e in envs:
        e.close()


def test_failed_walk_falls_back_to_plain_buffers(monkeypatch):
    """a zone walk that fails (another process took the memory meanwhile -> PGX_E_NOMEM) must not surface
    as an error of step(): the buffers then come from torch's allocator and `placement` says why."""
    import torch
    from pogema_amd import GridConfig, VecPogema, _lib
    gc = GridConfig(size=16, num_agents=8, obs_radius=3, density=0.2, seed=1)
    env = VecPogema(gc, batch=64, auto_reset=True, reuse_buffers=True, placement_budget_gib=4.0)
    env.PLACEMENT_MIN_BYTES = 1  # force the pool path for this small tensor
    calls = []

    def boom(count, skip_gib=0.0):
        calls.append(count)
        raise _lib.PgxError(-3, "hipMemCreate/hipMemMap (second half): out of memory")

    monkeypatch.setattr(env, "_zone_pool", boom)
    env.reset(seed=1)
    acts = torch.randint(0, 5, (64, 8), device="cuda", dtype=torch.int8)
    obs, *_ = env.step(acts)
    assert calls and env.placement["method"] == "torch allocator" and "out of memory" in env.placement["fallback"]
    assert obs.shape == env.obs_shape and not env.placement["spread"]
    out = env.rollout(steps=3, obs_slots=2)  # the rollout ring takes the same fallback
    assert out["obs"].shape[0] == 2 and out["obs"].is_contiguous()
    env.close()


def test_warm_buffers_is_explicit_and_capture_is_guarded():
    """warm_buffers() picks the output buffers at a moment of the caller's choosing; an un-warmed first step inside a
    graph capture is refused with a clear message instead of synchronising inside the capture."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=16, num_agents=8, obs_radius=3, density=0.2, seed=1)
    acts = torch.randint(0, 5, (64, 8), device="cuda", dtype=torch.int8)
    cold = VecPogema(gc, batch=64, auto_reset=True, reuse_buffers=True)
    cold.reset(seed=1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="warm_buffers"):
            with torch.cuda.graph(g, stream=s):
                cold.step(acts)
    torch.cuda.synchronize()
    cold.close()
    env = VecPogema(gc, batch=64, auto_reset=True, reuse_buffers=True)
    env.reset(seed=1)
    assert env._bufs is None
    pl = env.warm_buffers()
    assert env._bufs is not None and pl["method"] == "torch allocator"  # 0.3 MB: no walk for such a tensor
    bufs = env._bufs
    assert env.warm_buffers() is env.placement and env._bufs is bufs  # idempotent
    ref = VecPogema(gc, batch=64, auto_reset=True)
    ref.reset(seed=1)
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o1 = env.step(acts)[0]
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(o1, ref.step(acts)[0])
    env.close(); ref.close()


def test_address_space_accounting():
    """Pools never give their address ranges back nor re-use them (ROCm 7.2 keeps stale translations either way:
    profiles/r2/vmm_va_reuse_fault.txt, profiles/r3/vmm_va_remap_stale.txt), so every pool costs count x stride of
    address space -- and nothing more: the probe chunks of the zone walk are plain hipMalloc memory."""
    import torch
    from pogema_amd import _lib
    from pogema_amd.buffers import ZoneBuffers
    lib = _lib.load()
    ptrs = []
    for shape, kw in (((5, 1 << 20), dict(max_spacer_gib=0)), ((5, 1 << 20), dict(max_spacer_gib=0)),
                      ((3072, 64, 3, 11, 11), dict())):  # the last one walks (285 MB per buffer)
        before = int(lib.pgx_buffers_va_reserved())
        pool = ZoneBuffers(shape, torch.float32, "cuda:0", count=2, **kw)
        assert int(lib.pgx_buffers_va_reserved()) - before == 2 * pool.stride_bytes
        for i, t in enumerate(pool.tensors):
            t.fill_(float(i + 1))
        torch.cuda.synchronize()
        assert all(bool((t == float(i + 1)).all()) for i, t in enumerate(pool.tensors))
        ptrs.append(pool.tensors[0].data_ptr())
        del pool, t
    assert len(set(ptrs)) == 3, "a released range is not handed out again"


def test_recycle_mode_never_aliases_and_takes_sets_back():
    """reuse_buffers='recycle' (the default): outputs come from a few recycled sets (zone-spread observation buffers for
    large tensors), a set is handed out again only when the caller has dropped every reference to every member (views
    included), and while all are held the engine falls back to fresh tensors -- results identical to
    reuse_buffers=False throughout."""
    import gc as pygc
    import torch
    from pogema_amd import GridConfig, VecPogema
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=3, collision_system="soft")
    B = 512
    ref = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False)
    env = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib=8.0)  # default mode; the walk asked for explicitly
    assert env.recycle and not env.reuse_buffers
    env.PLACEMENT_MIN_BYTES = 1 << 20  # 23.8 MB per tensor: force the pool path
    ref.reset(seed=3)
    first, _ = env.reset(seed=3)
    rec = env._recycler
    assert rec and len(rec) == 3 and env.placement["method"].startswith("pgx_buffers")
    pool_ptrs = set(rec.obs_pointers())
    assert first.data_ptr() in pool_ptrs
    acts = torch.randint(0, 5, (9, B, 32), device="cuda", dtype=torch.int8)
    held, expect = [first], [ref.observe()]
    for t in range(5):  # hold everything: 3 sets, then fresh tensors; nothing may be overwritten
        held.append(env.step(acts[t])[0])
        expect.append(ref.step(acts[t])[0])
    assert len({h.data_ptr() for h in held}) == len(held)
    assert sum(h.data_ptr() in pool_ptrs for h in held) == len(rec) and rec.misses == len(held) - len(rec)
    for h, e in zip(held, expect):
        assert torch.equal(h, e)
    view = held[1][3, 2]           # a view keeps its set out of circulation
    kept_ptr = held[1].data_ptr()
    del held, h, first
    pygc.collect()
    assert rec.free_sets() == len(rec) - 1
    got = [env.step(acts[5])[0], env.step(acts[6])[0], env.step(acts[7])[0]]
    want = [ref.step(acts[5])[0], ref.step(acts[6])[0], ref.step(acts[7])[0]]
    assert kept_ptr not in {g.data_ptr() for g in got} and all(torch.equal(g, w) for g, w in zip(got, want))
    assert torch.equal(view, expect[1][3, 2])
    # a small member alone (rewards) keeps its set out as well
    del got, view
    pygc.collect()
    assert rec.free_sets() == len(rec)
    _, rewards, *_ = env.step(acts[8])
    want_r = ref.step(acts[8])[1]
    assert rec.free_sets() == len(rec) - 1 and torch.equal(rewards, want_r)
    # the memory outlives the environment for as long as a tensor references it
    env.close()
    del env, rec
    pygc.collect()
    assert torch.equal(rewards, want_r)
    ref.close()


def test_recycle_mode_small_tensors_and_missing_hook(monkeypatch):
    """Small observation tensors are recycled from torch's own memory (no zone walk); without the storage-count hook the
    mode degrades to fresh tensors per step."""
    import torch
    from pogema_amd import GridConfig, VecPogema, buffers
    cfg = GridConfig(size=16, num_agents=8, obs_radius=3, density=0.2, seed=1)
    acts = torch.randint(0, 5, (64, 8), device="cuda", dtype=torch.int8)
    env = VecPogema(cfg, batch=64, auto_reset=True)
    env.reset(seed=1)
    assert env._recycler and env.placement["method"] == "torch allocator"
    ptrs = set()
    for _ in range(12):
        obs = env.step(acts)[0]
        ptrs.add(obs.data_ptr())
    assert len(ptrs) <= len(env._recycler) and env._recycler.misses == 0
    env.close()
    monkeypatch.setattr(buffers.RecyclingOutputs, "available", staticmethod(lambda: False))
    env = VecPogema(cfg, batch=64, auto_reset=True)
    env.reset(seed=1)
    assert env._recycler is False
    a = env.step(acts)[0]
    b = env.step(acts)[0]
    assert a.data_ptr() != b.data_ptr()
    env.close()


def test_default_policy_walks_only_on_a_device_that_is_ours(monkeypatch):
    """placement_budget_gib=None (the default) must not hold half of a SHARED device.  The walk runs only when
    >= 90 % of the device's memory is free and nobody else is walking it; otherwise NOTHING is held: one probe pair tells
    whether the allocator happens to stand between two zones (then the n buffers are built there), no spacers, no spare
    buffers, no timing tensors, no cache flush, no device-wide synchronisation -- and `placement["policy"]` says so."""
    import torch
    from pogema_amd import GridConfig, VecPogema, buffers
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=5)
    B = 3000  # 139 MB per observation tensor: above the walk threshold
    acts = torch.randint(0, 5, (B, 32), device="cuda", dtype=torch.int8)
    real = torch.cuda.mem_get_info
    total = real(0)[1]
    walks = []
    orig_pool = VecPogema._zone_pool

    def counting_pool(self, count, skip_gib=0.0):
        walks.append(self._budget_now)
        return orig_pool(self, count, skip_gib)

    monkeypatch.setattr(VecPogema, "_zone_pool", counting_pool)
    flushes, syncs, probes = [], [], []
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: flushes.append(1))
    real_sync = torch.cuda.synchronize
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: (syncs.append(1), real_sync(*a, **k))[1])
    real_zone = buffers.ZoneBuffers.__init__

    def spying_zone(self, shape, dtype, device, count=2, max_spacer_gib=None, skip_gib=0.0, sync_device=True):
        probes.append((count, max_spacer_gib, sync_device))
        return real_zone(self, shape, dtype, device, count=count, max_spacer_gib=max_spacer_gib, skip_gib=skip_gib,
                         sync_device=sync_device)

    monkeypatch.setattr(buffers.ZoneBuffers, "__init__", spying_zone)
    # (1) a loaded / shared device: 60 % free
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(0.6 * total), total))
    env = VecPogema(cfg, batch=B, auto_reset=True)
    syncs_before = len(syncs)
    env.reset(seed=5)  # (the output buffers are picked here: the first observation needs one)
    obs = env.step(acts)[0]
    # exactly the 2 buffers it needs, a probe-only budget, no device-wide synchronisation, no walk, no flush
    assert probes == [(2, VecPogema.PROBE_ONLY_GIB, False)] and not walks and not flushes and len(syncs) == syncs_before
    pl = env.placement
    assert pl["policy"].startswith("probe only, nothing held") and "shared or already loaded" in pl["policy"]
    assert pl["candidates"] == 0 and (pl.get("spacer_gib") or 0.0) == 0.0
    assert pl["method"] == ("pgx_buffers (two HBM zones per buffer)" if pl["spread"] else "torch allocator")
    assert env._recycler and len(env._recycler) == 2 and obs.data_ptr() in set(env._recycler.obs_pointers())
    ref = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False, placement_budget_gib=0)
    ref.reset(seed=5)
    assert torch.equal(ref.step(acts)[0], obs)
    ref.close()
    env.close(release=True)  # (a zone set found by the probe would otherwise be taken over by the next case without a walk)
    del probes[:]
    # (2) the same, but the caller asks: explicit budgets walk wherever they are
    env = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib=2.0)
    env.reset(seed=5)
    assert walks and walks[0] == 2.0 and env.placement["policy"].startswith("explicit") and env.placement["budget_gib"] == 2.0
    env.close(release=True)
    assert buffers.ParkedBuffers.bytes_parked() == 0
    # (3) an empty device, but another process holds the walk lock: it is not ours alone -> no walk, and no waiting
    del walks[:]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(0.97 * total), total))
    with buffers.walk_lock(0) as held:
        assert held
        env = VecPogema(cfg, batch=B, auto_reset=True)
        env.reset(seed=5)
        assert not walks and "another process is walking" in env.placement["policy"]
        assert env.placement["policy"].startswith("probe only, nothing held") and probes[-1] == (2, VecPogema.PROBE_ONLY_GIB, False)
        env.close(release=True)
    # (4) an empty device and nobody else: the walk runs with half of the free memory
    env = VecPogema(cfg, batch=B, auto_reset=True)
    env.reset(seed=5)
    assert walks and abs(walks[0] - 0.5 * 0.97 * total / 2 ** 30) < 1.0 and env.placement["policy"].startswith("auto")
    assert env.placement["method"].startswith("pgx_buffers")
    env.close(release=True)
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)


@pytest.mark.parametrize("budget", [0, 1.0, "half", "all"])
def test_placement_budget_parameter(budget):
    """VecPogema(placement_budget_gib=...): 0 = never walk (plain torch buffers), a number = GiB the walk may hold, "half" =
    half of the free memory, "all" = everything but the engine's reserve; results are the same tensors either way."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=5, collision_system="priority")
    B = 3000  # 139 MB per observation tensor: the pool path without an override
    env = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib=budget)
    ref = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False)
    o, _ = env.reset(seed=5)
    r, _ = ref.reset(seed=5)
    assert torch.equal(o, r)
    pl = env.placement
    assert pl["policy"].startswith("explicit")
    if budget == 0:
        assert pl["method"] == "torch allocator" and pl["budget_gib"] == 0 and pl["candidates"] == 0 and not pl["spread"]
    else:
        assert pl["method"].startswith("pgx_buffers")
    if budget == 1.0:
        assert pl["budget_gib"] == 1.0 and pl["spacer_gib"] <= 1.0
    with pytest.raises(ValueError):
        VecPogema(cfg, batch=4, placement_budget_gib="lots")
    acts = torch.randint(0, 5, (3, B, 32), device="cuda", dtype=torch.int8)
    for t in range(3):
        a, b = env.step(acts[t]), ref.step(acts[t])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    env.close(); ref.close()


def test_closed_environments_leave_their_buffers_to_the_next_one(monkeypatch):
    """buffers.ParkedBuffers: close() parks the zone-spread observation buffers nobody references any more -- as a WHOLE
    set or not at all --; the next environment with the same observation tensor takes them over without a
    walk (and without new address space); a set the caller still holds a buffer of is NOT parked (the shelf never holds
    an unclaimable remainder); PGX_POOL_CACHE_MB=0 switches the shelf off; close(release=True) empties it."""
    import gc as pygc
    import torch
    from pogema_amd import GridConfig, VecPogema, _lib
    from pogema_amd.buffers import ParkedBuffers
    lib = _lib.load()
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=3, collision_system="soft")
    B = 3000  # 139 MB per observation tensor -> two output sets from the pool
    acts = torch.randint(0, 5, (4, B, 32), device="cuda", dtype=torch.int8)
    ref = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False)
    ref.reset(seed=3)
    want = [ref.step(acts[t])[0] for t in range(4)]
    key = (0, (B, 32, 3, 11, 11), torch.float32)

    a = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib="half")
    a.reset(seed=3)
    assert a.placement["method"] == "pgx_buffers (two HBM zones per buffer)" and len(a._recycler) == 2
    ptrs_a = set(a._recycler.obs_pointers())
    assert torch.equal(a.step(acts[0])[0], want[0])
    a.close()
    pygc.collect()
    if len(a._zone_ptrs) == 2:  # (the candidate timing may prefer one of torch's own buffers: then no whole zone set exists)
        assert len(ParkedBuffers._shelf[key]) == 1 and len(ParkedBuffers._shelf[key][0][0]) == 2
        assert ParkedBuffers.bytes_parked() == 2 * B * 32 * 3 * 121 * 4
    else:
        pytest.skip("the pool's buffers lost the candidate timing on this box: nothing to park")

    va = int(lib.pgx_buffers_va_reserved())
    b = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib="half")
    b.reset(seed=3)
    assert "taken over from a closed environment" in b.placement["method"] and key not in ParkedBuffers._shelf
    assert set(b._recycler.obs_pointers()) == ptrs_a and int(lib.pgx_buffers_va_reserved()) == va
    for t in range(4):
        assert torch.equal(b.step(acts[t])[0], want[t])
    held = b.step(acts[0])[0]  # the caller keeps this one: its buffer must not be parked
    b.close()
    assert key not in ParkedBuffers._shelf and ParkedBuffers.bytes_parked() == 0  # one of two: not a set -> nothing is kept

    from pogema_amd.buffers import WalkVerdicts
    WalkVerdicts.clear()  # (on a lease where `a`'s walk found no second zone the negative cache would send `c` to probe-only)
    c = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib="half")  # nothing parked: its own walk
    c.reset(seed=3)
    assert c.placement["method"] == "pgx_buffers (two HBM zones per buffer)"
    assert held.data_ptr() not in set(c._recycler.obs_pointers())
    monkeypatch.setenv("PGX_POOL_CACHE_MB", "0")
    c.close()
    assert ParkedBuffers.bytes_parked() == 0  # switched off: nothing added
    monkeypatch.delenv("PGX_POOL_CACHE_MB")
    WalkVerdicts.clear()
    d = VecPogema(cfg, batch=B, auto_reset=True, placement_budget_gib="half")
    d.reset(seed=3)
    whole = len(d._zone_ptrs) == 2
    d.close()
    assert (ParkedBuffers.bytes_parked() > 0) == whole
    e = VecPogema(cfg, batch=4, auto_reset=True)
    e.close(release=True)  # ... gives back what earlier environments left behind, too
    assert ParkedBuffers.bytes_parked() == 0
    from pogema_amd import release_cached_buffers
    release_cached_buffers()
    ref.close()


def test_walk_lock_names_the_physical_device():
    """The per-device walk lock must be keyed by something processes with different HIP_VISIBLE_DEVICES agree on (8 ranks
    that all call their GPU 'device 0' must not share one lock, two processes on one GPU must)."""
    from pogema_amd.buffers import device_identity
    ident = device_identity(0)
    assert ident.startswith(("uuid_", "pci_")), f"no physical identity for device 0: {ident!r}"


def test_a_failed_walk_is_paid_once_per_process(monkeypatch):
    """VERDICT r4 #1b: once a full-budget walk has found no second zone on a device, later engines of the process -- of any
    shape, also rollout rings -- go straight to the probe-only placement (nothing held) instead of holding the memory
    again.  The failure is forced: PGX_ZONE_SCAN makes the walk run its whole budget and accept nothing."""
    import torch
    from pogema_amd import GridConfig, VecPogema, buffers, release_cached_buffers
    monkeypatch.setenv("PGX_ZONE_SCAN", "1")
    monkeypatch.setenv("PGX_ZONE_SPACER_GIB", "24")
    release_cached_buffers()
    w0 = buffers.WalkVerdicts.walks
    probes = []
    real_zone = buffers.ZoneBuffers.__init__

    def spying_zone(self, shape, dtype, device, count=2, max_spacer_gib=None, skip_gib=0.0, sync_device=True):
        probes.append((count, max_spacer_gib))
        return real_zone(self, shape, dtype, device, count=count, max_spacer_gib=max_spacer_gib, skip_gib=skip_gib,
                         sync_device=sync_device)

    monkeypatch.setattr(buffers.ZoneBuffers, "__init__", spying_zone)
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=5)
    a = VecPogema(cfg, batch=3000, auto_reset=True)
    a.reset(seed=5)
    assert buffers.WalkVerdicts.walks == w0 + 1 and not a.placement["spread"] and a.placement["candidates"] == 3
    assert buffers.WalkVerdicts.failed(0, 24.0) is not None and buffers.WalkVerdicts.failed(0, 100.0) is None
    n_walk_probes = len(probes)
    assert probes[0][1] == 24.0
    # another shape, same process: no walk, a probe-only pool of exactly the buffers it needs
    b = VecPogema(cfg, batch=3300, auto_reset=True)
    b.reset(seed=5)
    assert buffers.WalkVerdicts.walks == w0 + 1, "the failed walk was repeated"
    assert "negative cache" in b.placement["policy"] and b.placement["policy"].startswith("probe only, nothing held")
    assert probes[n_walk_probes:] == [(2, VecPogema.PROBE_ONLY_GIB)]
    assert b.placement.get("same_zone_us", 0) > 0, "the probe's timing must survive into the placement record (box_store_stream_gbs)"
    # ... and a rollout ring of a third engine: exactly its slots, no timing pass
    del probes[:]
    c = VecPogema(cfg, batch=3100, auto_reset=True)
    c.reset(seed=5)
    acts = torch.randint(0, 5, (4, 3100, 32), device="cuda", dtype=torch.int8)
    ref = VecPogema(cfg, batch=3100, auto_reset=True, reuse_buffers=False, placement_budget_gib=0)
    ref.reset(seed=5)
    out = c.rollout(acts, obs_slots=2)
    want = [ref.step(acts[t])[0] for t in range(4)]
    assert torch.equal(out["obs"][0], want[2]) and torch.equal(out["obs"][1], want[3])
    assert buffers.WalkVerdicts.walks == w0 + 1 and all(p[1] == VecPogema.PROBE_ONLY_GIB for p in probes)
    assert (2, VecPogema.PROBE_ONLY_GIB) in probes
    # a larger explicit budget than the one that failed may walk again; forgetting the verdict re-enables the default
    d = VecPogema(cfg, batch=3200, auto_reset=True, placement_budget_gib=48.0)
    d.reset(seed=5)
    assert buffers.WalkVerdicts.walks == w0 + 2 and d.placement["candidates"] == 6
    release_cached_buffers()
    assert buffers.WalkVerdicts.failed(0, 24.0) is None
    monkeypatch.setenv("PGX_WALK_NEGATIVE_CACHE", "0")
    buffers.WalkVerdicts.note_walk(0, {"spread": False, "candidates": 3}, 24.0)
    assert buffers.WalkVerdicts.failed(0, 24.0) is None
    for e in (a, b, c, d, ref):
        e.close(release=True)


def test_explicit_budget_below_one_spacer_is_refused():
    """placement_budget_gib in (0, 1) used to become the probe-only mode silently while the policy text still
    said 'explicit'."""
    from pogema_amd import GridConfig, VecPogema
    with pytest.raises(ValueError, match="cannot hold a single spacer"):
        VecPogema(GridConfig(size=8, num_agents=2), batch=4, placement_budget_gib=0.5)


def test_rollout_ring_probe_only_builds_exactly_its_slots(monkeypatch):
    """on a shared / loaded device (or with a busy walk lock) the rollout ring is built like the step buffers:
    exactly `slots` buffers, no timing pass, no drops (pgx_buffers_drop synchronises the device), nothing held."""
    import torch
    from pogema_amd import GridConfig, VecPogema, buffers
    cfg = GridConfig(size=32, num_agents=32, obs_radius=5, density=0.3, seed=5)
    B = 3000
    real = torch.cuda.mem_get_info
    total = real(0)[1]
    probes, drops, timed = [], [], []
    real_zone = buffers.ZoneBuffers.__init__

    def spying_zone(self, shape, dtype, device, count=2, max_spacer_gib=None, skip_gib=0.0, sync_device=True):
        probes.append((count, max_spacer_gib, sync_device))
        return real_zone(self, shape, dtype, device, count=count, max_spacer_gib=max_spacer_gib, skip_gib=skip_gib,
                         sync_device=sync_device)

    monkeypatch.setattr(buffers.ZoneBuffers, "__init__", spying_zone)
    monkeypatch.setattr(buffers.ZoneBuffers, "drop", lambda self, i: drops.append(i))
    real_time = VecPogema._time_observe
    monkeypatch.setattr(VecPogema, "_time_observe", lambda self, *a: (timed.append(1), real_time(self, *a))[1])
    acts = torch.randint(0, 5, (3, B, 32), device="cuda", dtype=torch.int8)
    ref = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False, placement_budget_gib=0)
    ref.reset(seed=5)
    want = [ref.step(acts[t])[0] for t in range(3)]
    for case in ("loaded", "locked"):
        del probes[:], drops[:], timed[:]
        env = VecPogema(cfg, batch=B, auto_reset=True, reuse_buffers=False)
        env.reset(seed=5)
        if case == "loaded":
            monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(0.6 * total), total))
            out = env.rollout(acts, obs_slots=3)
        else:
            monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (int(0.97 * total), total))
            with buffers.walk_lock(0) as held:
                assert held
                out = env.rollout(acts, obs_slots=3)
        assert probes == [(3, VecPogema.PROBE_ONLY_GIB, False)] and not drops and not timed, (case, probes, drops, timed)
        assert env.placement["policy"].startswith("probe only, nothing held")
        for t in range(3):
            assert torch.equal(out["obs"][t], want[t])
        env.close(release=True)
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)
    ref.close()


def test_rollout_ring_is_borrowed_from_the_recycler_sets():
    """rollout()'s observation ring of up to as many slots as the engine has output sets comes out of those
    sets -- placed and timed once -- and stays out of circulation for as long as the caller holds it; step() keeps working
    from the rest and never overwrites it."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, seed=0, collision_system="soft", max_episode_steps=16)
    env = VecPogema(gc, batch=4096, device="cuda:0", auto_reset=True)  # 380 MB observations: three recycled output sets
    env.reset(seed=0)
    acts = torch.randint(0, 5, (6, 4096, 64), device="cuda:0", dtype=torch.int8)
    env.step(acts[0])
    rec = env._recycler
    assert rec and len(rec) == 3 and rec.free_sets() == 3
    out = env.rollout(acts, obs_slots=2)
    ring = out["obs"]
    base = ring.data_ptr()
    assert base in rec.obs_pointers() and ring[1].data_ptr() in rec.obs_pointers(), "the ring's slots are two of the engine's own buffers"
    assert rec.free_sets() == 1 and not env._rollout_pools, "two sets are out, no ring of its own was built"
    keep = ring.clone()
    for t in range(4):  # step() serves from the remaining set (and fresh tensors while its result is held): the ring is untouched
        obs, *_ = env.step(acts[t])
        assert obs.data_ptr() not in (base, ring[1].data_ptr())
    assert torch.equal(ring, keep)
    del obs
    view = ring[0, :8]
    del out, ring
    assert rec.free_sets() <= 2, "a view keeps its set out"
    del view, keep
    assert rec.free_sets() == 3, "dropped: all sets are back"
    out3 = env.rollout(acts, obs_slots=3)   # three slots: only if the three buffers happen to lie at equal distances
    assert out3["obs"].shape[0] == 3
    ref = VecPogema(gc, batch=4096, device="cuda:0", auto_reset=True)
    ref.reset(seed=0)
    ref.step(acts[0])
    ref.rollout(acts, obs_slots=2)
    for t in range(4):
        ref.step(acts[t])
    for t in range(6):
        robs, *_ = ref.step(acts[t])
    assert torch.equal(out3["obs"][5 % 3], robs), "same trajectory whatever memory the ring lives in"
    env.close()
    ref.close()
