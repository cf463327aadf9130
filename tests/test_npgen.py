"""pgx_np_generate_host (the device kernel's code, compiled for the host) against the numpy-written restatement of
upstream's generator (oracle/generator_oracle.py: generate_instance_numpy).  What is pinned here: the numpy arithmetic
and the placing rule AS RECALLED; whether the recollection matches the real package is open (docs/SPEC.md)."""
import numpy as np
import pytest

from oracle import generator_oracle as G
from pogema_amd.nprng import np_generate_host

CASES = [(8, 8, 2, 0.3), (16, 16, 8, 0.3), (32, 32, 16, 0.3), (64, 64, 64, 0.3), (12, 9, 40, 0.1), (10, 10, 30, 0.6),
         (5, 5, 1, 0.0), (7, 13, 5, 0.45), (3, 3, 4, 0.0), (1, 9, 2, 0.2)]
SEEDS = [0, 1, 2, 7, 12345, 2 ** 40 + 3, 2 ** 63 + 11]


def _check(seeds, H, W, A, density, given_map=None):
    o, a, t, st = np_generate_host(seeds, H, W, A, density, given_map)
    placed = 0
    for i, s in enumerate(seeds):
        try:
            ro, ra, rt = G.generate_instance_numpy(int(s), H, W, A, density, given_map)
        except OverflowError:
            assert st[i] == 1
            continue
        assert st[i] == 0
        np.testing.assert_array_equal(o[i], ro)
        np.testing.assert_array_equal(a[i], ra)
        np.testing.assert_array_equal(t[i], rt)
        placed += 1
    return placed


@pytest.mark.parametrize("H,W,A,density", CASES)
def test_host_generator_equals_numpy(H, W, A, density):
    _check(SEEDS, H, W, A, density)


def test_instances_are_valid_and_reachable():
    H = W = 24
    A = 12
    o, a, t, st = np_generate_host(np.arange(40), H, W, A, 0.3)
    assert (st == 0).all()
    from oracle.pogema_oracle import label_components
    for b in range(40):
        labels, _ = label_components(o[b])
        assert (o[b][a[b, :, 0], a[b, :, 1]] == 0).all() and (o[b][t[b, :, 0], t[b, :, 1]] == 0).all()
        assert len({tuple(p) for p in a[b]}) == A and len({tuple(p) for p in t[b]}) == A
        assert (labels[a[b, :, 0], a[b, :, 1]] == labels[t[b, :, 0], t[b, :, 1]]).all()


def test_given_map_only_draws_positions():
    rng = np.random.default_rng(5)
    m = (rng.random((14, 11)) < 0.25).astype(np.uint8)
    assert _check(SEEDS, 14, 11, 6, 0.3, m) == len(SEEDS)
    o, *_ = np_generate_host([3], 14, 11, 6, 0.3, m)
    np.testing.assert_array_equal(o[0], m)


def test_overflow_status_and_arguments():
    o, a, t, st = np_generate_host([1, 2], 4, 4, 9, 0.0)  # 16 cells hold 8 pairs
    assert (st == 1).all()
    with pytest.raises(Exception):
        np_generate_host([1], 4, 4, 2, 1.5)
    with pytest.raises(ValueError):
        np_generate_host([1], 4, 4, 2, 0.3, np.zeros((3, 4)))
