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
