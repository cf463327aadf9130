"""CPU: the engine's host-side instance generator (pgx_generate / pgx_place_agents) -- invariants
of upstream pogema/generator.py's contract: starts and targets on distinct free cells, every pair
inside one 4-connected component; deterministic per seed; independent of thread count."""
import ctypes as C

import numpy as np
import pytest

from oracle.pogema_oracle import label_components
from pogema_amd import _lib
from util import generate_instances


@pytest.mark.parametrize("shape", [(6, 8, 8, 2, 0.3), (5, 16, 16, 8, 0.3), (3, 64, 64, 64, 0.3), (2, 9, 21, 30, 0.1)])
def test_instances_are_valid(shape):
    B, H, W, A, density = shape
    obstacles, agents, targets = generate_instances(B, H, W, A, density, 11)
    assert set(np.unique(obstacles)) <= {0, 1}
    for b in range(B):
        labels, _ = label_components(obstacles[b])
        cells = set()
        for i in range(A):
            s, t = tuple(agents[b, i]), tuple(targets[b, i])
            assert obstacles[b][s] == 0 and obstacles[b][t] == 0
            assert labels[s] == labels[t] >= 0, "start and target must share a component"
            assert s not in cells and t not in cells, "all 2A cells are distinct"
            cells.update((s, t))


def test_density_and_determinism():
    a = generate_instances(64, 32, 32, 4, 0.3, 5)
    b = generate_instances(64, 32, 32, 4, 0.3, 5)
    c = generate_instances(64, 32, 32, 4, 0.3, 6)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0], c[0])
    assert abs(a[0].mean() - 0.3) < 0.02
    # env i of seed s equals env 0 of seed s+i (per-env streams: sharding-independent)
    d = generate_instances(1, 32, 32, 4, 0.3, 5 + 7)
    assert np.array_equal(a[0][7], d[0][0]) and np.array_equal(a[1][7], d[1][0])


def test_thread_count_independent(engine_lib):
    outs = []
    for nt in (1, 3, 8):
        o = np.empty((40, 12, 12), np.uint8)
        a = np.empty((40, 6, 2), np.int32)
        t = np.empty((40, 6, 2), np.int32)
        _lib.check(engine_lib.pgx_generate(40, 12, 12, 6, C.c_float(0.25), 3, 10, nt, o.ctypes.data, a.ctypes.data, t.ctypes.data))
        outs.append((o, a, t))
    for o, a, t in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(a, outs[0][1]) and np.array_equal(t, outs[0][2])


def test_overflow_is_an_error(engine_lib):
    o = np.empty((1, 4, 4), np.uint8)
    a = np.empty((1, 9, 2), np.int32)
    t = np.empty((1, 9, 2), np.int32)
    status = engine_lib.pgx_generate(1, 4, 4, 9, C.c_float(0.0), 0, 3, 1, o.ctypes.data, a.ctypes.data, t.ctypes.data)
    assert status == -5 and "agents" in engine_lib.pgx_last_error().decode()
    status = engine_lib.pgx_generate(1, 4, 4, 4, C.c_float(0.9), 0, 3, 1, o.ctypes.data, a.ctypes.data, t.ctypes.data)
    assert status == -5


def test_place_on_given_map(engine_lib):
    m = np.zeros((5, 7), np.uint8)
    m[2, :] = 1  # wall splits the map in two components
    a = np.empty((8, 3, 2), np.int32)
    t = np.empty((8, 3, 2), np.int32)
    _lib.check(engine_lib.pgx_place_agents(8, 5, 7, 3, 42, 10, 2, m.ctypes.data, 1, a.ctypes.data, t.ctypes.data))
    assert ((a[..., 0] < 2) == (t[..., 0] < 2)).all(), "pairs stay on their side of the wall"
    assert (m[a[..., 0], a[..., 1]] == 0).all() and (m[t[..., 0], t[..., 1]] == 0).all()
