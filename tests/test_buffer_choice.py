"""The selection rule of VecPogema's output buffers (pogema_amd.vec_env.choose_buffers), on made-up timings."""
import pytest

torch = pytest.importorskip("torch")
from pogema_amd.vec_env import choose_buffers  # noqa: E402


def test_pool_buffers_win_ties_and_near_ties():
    assert choose_buffers([100.0, 101.0, 120.0], [100.0, 100.5], 2) == [("zone", 0), ("zone", 1)]
    assert choose_buffers([100.0, 101.0], [99.5, 99.9], 2) == [("zone", 0), ("zone", 1)]  # within 2 %


def test_clearly_faster_plain_buffers_replace_the_slowest_pool_buffer():
    assert choose_buffers([100.0, 130.0, 140.0], [110.0, 112.0], 2) == [("zone", 0), ("torch", 0)]
    assert sorted(choose_buffers([150.0, 151.0], [110.0, 112.0, 140.0], 2)) == [("torch", 0), ("torch", 1)]
    assert choose_buffers([146.0, 148.6, 150.0], [148.0], 2) == [("zone", 0), ("zone", 1)]  # a no-zone box: nothing to gain


def test_fewer_pool_buffers_than_needed_and_single_buffer_mode():
    assert choose_buffers([], [10.0, 11.0, 12.0], 2) == [("torch", 0), ("torch", 1)]
    assert choose_buffers([100.0], [90.0, 95.0], 2) == [("torch", 1), ("torch", 0)]
    assert choose_buffers([100.0, 105.0], [90.0], 1) == [("torch", 0)]
    assert choose_buffers([100.0, 105.0], [99.0], 1) == [("zone", 0)]


def test_recycle_set_count_by_tensor_size():
    """reuse_buffers='recycle': two output sets for 64-256 MiB observation tensors (two still share the Infinity Cache),
    three otherwise; BASELINE configs[1..4] = 11.9 / 761 / 190 / 2831 MB."""
    from pogema_amd.vec_env import VecPogema
    sizes = {"configs[1]": 1024 * 8 * 1452, "configs[2]": 8192 * 64 * 1452, "configs[3] shard": 8192 * 16 * 1452,
             "configs[4]": 4096 * 256 * 2700}
    assert {k: VecPogema._recycle_sets(v) for k, v in sizes.items()} == {
        "configs[1]": 3, "configs[2]": 3, "configs[3] shard": 2, "configs[4]": 3}


def test_recycling_outputs_on_cpu_tensors():
    """The recycler itself needs no GPU: sets are handed out least-recently-used first, a set comes back only when every
    member (and every view of it) has been dropped, and take() says None while all sets are out."""
    import gc
    import torch
    from pogema_amd.buffers import RecyclingOutputs
    assert RecyclingOutputs.available(), "see test_storage_count_hook_still_means_what_the_recycler_assumes"
    B, A = 4, 3
    masters = [torch.zeros((B, A, 3, 5, 5)) for _ in range(2)]
    rec = RecyclingOutputs(masters, B, A)
    del masters
    a = rec.take()
    b = rec.take()
    assert rec.take() is None and rec.misses == 1 and rec.free_sets() == 0
    assert a[0].data_ptr() != b[0].data_ptr() and a[1].shape == (B, A) and a[1].dtype == torch.float32
    assert all(t.dtype == torch.bool and t.shape == (B, A) for t in a[2:])
    a[1].fill_(7.0); a[2].fill_(True); a[3].fill_(False); a[4].fill_(True)  # members of a set do not overlap
    assert float(a[1].sum()) == 7.0 * B * A and bool(a[2].all()) and not bool(a[3].any()) and bool(a[4].all())
    first_ptr = a[0].data_ptr()
    keep = a[3][1]          # a view of a small member keeps the whole set out
    del a
    gc.collect()
    assert rec.free_sets() == 0 and rec.take() is None
    del keep
    gc.collect()
    assert rec.free_sets() == 1
    c = rec.take(with_obs=False)
    assert c[0] is None and rec.free_sets() == 0  # the small block is out; the set is busy although its obs buffer idles
    del b, c
    gc.collect()
    assert rec.free_sets() == 2
    d = rec.take()
    assert d[0].data_ptr() != first_ptr or len(rec) == 1  # least recently handed out first


def test_storage_count_hook_still_means_what_the_recycler_assumes():
    """reuse_buffers='recycle' reads a PRIVATE torch hook (torch._C._storage_Use_Count).  pogema_amd.buffers re-checks its
    meaning on a tiny CPU tensor before using it and degrades to fresh tensors (with a warning) when it changed; this
    test makes such a torch upgrade fail LOUDLY on the CPU instead of silently costing 20 % of the step time."""
    import warnings
    import pogema_amd.buffers as B
    B._HOOK = B._UNSET
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        hook = B.storage_count_hook()
    assert hook is not None, f"torch {torch.__version__}: the storage use-count hook is gone or changed its meaning"
    t = torch.zeros(4)
    c = t.untyped_storage()._cdata
    n = hook(c)
    u = t.view(2, 2)
    assert hook(c) == n + 1
    del u
    assert hook(c) == n


def test_a_changed_hook_degrades_with_a_warning(monkeypatch):
    import pogema_amd.buffers as B
    monkeypatch.setattr(B, "_HOOK", B._UNSET)
    monkeypatch.setattr(torch._C, "_storage_Use_Count", lambda cdata: 1)  # "changed its meaning"
    with pytest.warns(RuntimeWarning, match="no longer counts"):
        assert B.storage_count_hook() is None
    assert not B.RecyclingOutputs.available()
    monkeypatch.setattr(B, "_HOOK", B._UNSET)
    monkeypatch.delattr(torch._C, "_storage_Use_Count")
    with pytest.warns(RuntimeWarning, match="missing"):
        assert B.storage_count_hook() is None
    monkeypatch.undo()
    B._HOOK = B._UNSET
    assert B.storage_count_hook() is not None


def test_parked_buffers_keep_whole_sets_only(monkeypatch):
    """the shelf of closed environments' buffers holds only claimable sets, evicts whole sets, refuses sets
    above the limit and partial returns, and clear() empties it."""
    from pogema_amd.buffers import ParkedBuffers as P
    P.clear()
    monkeypatch.setenv("PGX_POOL_CACHE_MB", "3")
    mb = lambda n=1: torch.zeros(n << 20, dtype=torch.uint8)  # noqa: E731
    assert P.park("a", [mb(), mb()], {"spread": True}, 3) is False and P.bytes_parked() == 0   # fewer than a set: nothing kept
    assert P.park("a", [mb(), mb(), mb(), mb()], {"spread": True}, 3) is True                   # surplus buffer released
    assert P.bytes_parked() == 3 << 20
    assert P.claim("a", 2) is None                       # a set of another size is not claimable as such
    assert P.park("b", [mb(2), mb(2)], {}, 2) is False   # one set alone above the limit: not parked, "a" untouched
    assert P.bytes_parked() == 3 << 20
    assert P.park("c", [mb()], {"spread": False}, 1) is True   # limit exceeded: the oldest WHOLE set goes
    assert P.bytes_parked() == 1 << 20 and P.claim("a", 3) is None
    got = P.claim("c", 1)
    assert got is not None and len(got[0]) == 1 and got[1] == {"spread": False} and P.bytes_parked() == 0
    monkeypatch.setenv("PGX_POOL_CACHE_MB", "0")
    assert P.park("d", [mb()], {}, 1) is False           # shelf switched off
    monkeypatch.setenv("PGX_POOL_CACHE_MB", "8")
    assert P.park("d", [mb()], {}, 1) and P.park("d", [mb()], {}, 1) and P.bytes_parked() == 2 << 20
    P.clear()
    assert P.bytes_parked() == 0 and P.claim("d", 1) is None


def test_walk_lock_admits_one_walker_per_device():
    """The default policy skips the zone walk when another process (here: another open file description) is walking the
    same device; an explicit budget waits instead (not exercised: it would block)."""
    from pogema_amd.buffers import walk_lock
    with walk_lock(0) as first:
        assert first is True
        with walk_lock(0) as second:
            assert second is False, "a second concurrent walker of the same device must be turned away"
        with walk_lock(1) as other_device:
            assert other_device is True
    with walk_lock(0) as again:
        assert again is True, "the lock is released when the walk ends"


def test_walk_verdicts_negative_cache_logic(monkeypatch):
    """buffers.WalkVerdicts on the CPU: a failed full-budget walk is remembered per device, only for budgets it covers; a
    later success or clear() forgets it; PGX_WALK_NEGATIVE_CACHE=0 switches the cache off."""
    from pogema_amd.buffers import WalkVerdicts
    WalkVerdicts.clear()
    w0, f0 = WalkVerdicts.walks, WalkVerdicts.failed_walks
    assert WalkVerdicts.failed(0, 100.0) is None
    WalkVerdicts.note_walk(0, {"spread": False, "candidates": 17, "same_zone_us": 143.0, "spacer_gib": 136.0}, 143.4)
    v = WalkVerdicts.failed(0, 143.4)
    assert v and v["candidates"] == 17 and v["walks"] == 1 and WalkVerdicts.failed(0, 64.0) is not None
    assert WalkVerdicts.failed(0, 150.0) is not None, "a budget within one spacer of the failed one reaches nothing new"
    assert WalkVerdicts.failed(0, 272.0) is None, "a clearly larger budget may walk again"
    assert WalkVerdicts.failed(1, 10.0) is None, "verdicts are per device"
    WalkVerdicts.note_walk(0, {"spread": False, "candidates": 30}, 272.0)
    assert WalkVerdicts.failed(0, 272.0)["walks"] == 2
    assert (WalkVerdicts.walks, WalkVerdicts.failed_walks) == (w0 + 2, f0 + 2)
    monkeypatch.setenv("PGX_WALK_NEGATIVE_CACHE", "0")
    assert WalkVerdicts.failed(0, 10.0) is None
    monkeypatch.delenv("PGX_WALK_NEGATIVE_CACHE")
    WalkVerdicts.note_walk(0, {"spread": True, "candidates": 2}, 143.4)   # a walk that found a zone after all
    assert WalkVerdicts.failed(0, 10.0) is None and WalkVerdicts.failed_walks == f0 + 2
    WalkVerdicts.note_walk(0, {"spread": False, "candidates": 1}, 8.0)
    from pogema_amd import release_cached_buffers
    release_cached_buffers()
    assert WalkVerdicts.failed(0, 8.0) is None


def test_walk_lock_file_handling(tmp_path, monkeypatch):
    """the per-device lock file lives in a shared temp dir -- a planted symlink is not followed, another
    tenant's read-only file still locks, an unusable file means 'not ours' under the default policy (no walk) and
    'walk unlocked' only for an explicit request; two holders exclude each other."""
    import stat
    import tempfile
    from pogema_amd import buffers
    monkeypatch.setattr(tempfile, "gettempdir", lambda: str(tmp_path))
    monkeypatch.setattr(buffers, "device_identity", lambda i: f"testdev{i}")
    path = tmp_path / "pgx_zone_walk_testdev0.lock"
    with buffers.walk_lock(0) as a:
        assert a and path.exists() and stat.S_IMODE(path.stat().st_mode) == 0o666
        with buffers.walk_lock(0) as b:
            assert b is False, "a second walker on the same device must see the lock busy"
        with buffers.walk_lock(1) as c:
            assert c, "another device has its own lock"
    with buffers.walk_lock(0) as again:
        assert again
    # read-only file of 'another tenant': flock works on a read-only descriptor
    path.chmod(0o444)
    with buffers.walk_lock(0) as ro:
        assert ro
    # a symlink planted under the predictable name is not followed
    path.unlink()
    victim = tmp_path / "victim"
    victim.write_text("precious")
    path.symlink_to(victim)
    with buffers.walk_lock(0, wait=False) as planted:
        assert planted is False, "unusable lock file: the default policy must not walk"
    with buffers.walk_lock(0, wait=True) as explicit:
        assert explicit is True, "an explicit budget walks as asked, unlocked"
    assert victim.read_text() == "precious"


def test_walk_verdict_remembers_what_the_walk_covered_and_expires(monkeypatch):
    """a walk that was cut short (low free memory, a co-tenant) only vouches for the stretch it covered; a verdict
    ages out (TTL) and is dropped when the device has materially more free memory than at the failed walk."""
    from pogema_amd.buffers import WalkVerdicts
    WalkVerdicts.clear()
    free = {"gib": 50.0}
    monkeypatch.setattr(WalkVerdicts, "_free_gib", staticmethod(lambda idx: free["gib"]))
    # asked for 136 GiB, but the walk ended after 40 GiB of spacers
    WalkVerdicts.note_walk(0, {"spread": False, "candidates": 5, "spacer_gib": 40.0}, 136.0)
    v = WalkVerdicts.failed(0, 40.0)
    assert v is not None and v["budget_gib"] == 48.0 and v["requested_gib"] == 136.0 and v["free_gib"] == 50.0
    assert WalkVerdicts.failed(0, 56.0) is not None, "within one spacer of what was covered"
    assert WalkVerdicts.failed(0, 136.0) is None, "the stretch beyond 48 GiB was never walked: a full budget walks again"
    # more memory has become free since: the verdict is void
    free["gib"] = 59.0
    assert WalkVerdicts.failed(0, 40.0) is None
    free["gib"] = 50.0
    WalkVerdicts.note_walk(0, {"spread": False, "candidates": 5, "spacer_gib": 40.0}, 136.0)
    assert WalkVerdicts.failed(0, 40.0) is not None
    monkeypatch.setattr(WalkVerdicts, "TTL_S", 0.0)
    assert WalkVerdicts.failed(0, 40.0) is None, "older than the TTL"
    WalkVerdicts.clear()


def test_standin_pin_is_refused_at_the_default_location(tmp_path, monkeypatch):
    """a pin file derived from the stand-in package must never become the product's silent default; named
    explicitly (the rehearsal) it is the caller's business.  The parsed file is cached by (path, mtime, size)."""
    import json
    from pogema_amd import semantics as S
    pin = tmp_path / "pinned_semantics.json"
    pin.write_text(json.dumps({"switches": {"soft_vertex": "all_stay"}, "standin": True}))
    monkeypatch.delenv("PGX_SEMANTICS", raising=False)
    monkeypatch.delenv("PGX_PINNED_SEMANTICS_FILE", raising=False)
    monkeypatch.setattr(S, "pinned_file", lambda: str(pin))  # = "it lies where the product looks by default"
    with pytest.raises(ValueError, match="stand-in"):
        S.Semantics.from_env()
    monkeypatch.setenv("PGX_PINNED_SEMANTICS_FILE", str(pin))
    assert S.Semantics.from_env().soft_vertex == "all_stay"
    assert str(pin) in S._PIN_CACHE and S.pinned_defaults() == {"soft_vertex": "all_stay"}
    pin.write_text(json.dumps({"switches": {"soft_vertex": "lowest_index", "coop_reward": "per_agent"}, "standin": False}))
    assert S.pinned_defaults() == {"soft_vertex": "lowest_index", "coop_reward": "per_agent"}, "a rewritten file is re-read"
    # tests/pin_semantics.py refuses to write a stand-in pin into pogema_amd/
    import os
    import subprocess
    import sys
    d = tmp_path / "fx"
    d.mkdir()
    (d / "reference_probes.json").write_text(json.dumps({"standin": True}))
    import numpy as np
    np.savez(d / "reference_dummy.npz", x=np.zeros(1))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    target = os.path.join(root, "pogema_amd", "pinned_semantics_test_refusal.json")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "pin_semantics.py"), str(d), "--write-pin", target],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "STAND-IN" in (p.stderr + p.stdout) and not os.path.exists(target)
