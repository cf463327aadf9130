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
    if not RecyclingOutputs.available():
        import pytest
        pytest.skip("torch._C._storage_Use_Count missing in this torch build")
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
