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
    # global env i of seed s is the same instance wherever its shard starts (per-env streams: sharding-independent)
    d = generate_instances(1, 32, 32, 4, 0.3, 5, env_index_base=7)
    assert np.array_equal(a[0][7], d[0][0]) and np.array_equal(a[1][7], d[1][0])
    # seed and env index are separate key components: adjacent seeds share NO instance (a train/eval split by seed
    # must not see shifted copies of the same maps)
    maps_a = {m.tobytes() for m in a[0]}
    assert not any(m.tobytes() in maps_a for m in c[0])


def test_thread_count_independent(engine_lib):
    outs = []
    for nt in (1, 3, 8):
        o = np.empty((40, 12, 12), np.uint8)
        a = np.empty((40, 6, 2), np.int32)
        t = np.empty((40, 6, 2), np.int32)
        _lib.check(engine_lib.pgx_generate(40, 12, 12, 6, C.c_float(0.25), 3, 0, 10, nt, o.ctypes.data, a.ctypes.data, t.ctypes.data))
        outs.append((o, a, t))
    for o, a, t in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(a, outs[0][1]) and np.array_equal(t, outs[0][2])


def test_overflow_is_an_error(engine_lib):
    o = np.empty((1, 4, 4), np.uint8)
    a = np.empty((1, 9, 2), np.int32)
    t = np.empty((1, 9, 2), np.int32)
    status = engine_lib.pgx_generate(1, 4, 4, 9, C.c_float(0.0), 0, 0, 3, 1, o.ctypes.data, a.ctypes.data, t.ctypes.data)
    assert status == -5 and "agents" in engine_lib.pgx_last_error().decode()
    status = engine_lib.pgx_generate(1, 4, 4, 4, C.c_float(0.9), 0, 0, 3, 1, o.ctypes.data, a.ctypes.data, t.ctypes.data)
    assert status == -5


def test_place_on_given_map(engine_lib):
    m = np.zeros((5, 7), np.uint8)
    m[2, :] = 1  # wall splits the map in two components
    a = np.empty((8, 3, 2), np.int32)
    t = np.empty((8, 3, 2), np.int32)
    _lib.check(engine_lib.pgx_place_agents(8, 5, 7, 3, 42, 0, 10, 2, m.ctypes.data, 1, a.ctypes.data, t.ctypes.data))
    assert ((a[..., 0] < 2) == (t[..., 0] < 2)).all(), "pairs stay on their side of the wall"
    assert (m[a[..., 0], a[..., 1]] == 0).all() and (m[t[..., 0], t[..., 1]] == 0).all()


# ---- generator "GEN v2": normative Python statement == plain-C port == the product's host generator ----
GEN_CASES = [(6, 8, 8, 2, 0.3, 11), (5, 16, 16, 8, 0.3, 3), (3, 9, 21, 30, 0.1, 77), (2, 32, 32, 16, 0.45, 5)]


def _c_generate(B, H, W, A, density, seed, base=0, given_map=None, epochs=None, max_retries=50):
    from oracle.c_oracle import load
    lib = load()
    lib.po_generate.argtypes = [C.c_int32] * 4 + [C.c_float, C.c_uint64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.po_generate.restype = C.c_int
    o = np.ascontiguousarray(given_map, np.uint8) if given_map is not None else np.empty((B, H, W), np.uint8)
    a = np.empty((B, A, 2), np.int32)
    t = np.empty((B, A, 2), np.int32)
    ep = None if epochs is None else np.ascontiguousarray(epochs, np.uint32).ctypes.data
    st = lib.po_generate(B, H, W, A, density, seed, base, ep, max_retries, int(given_map is not None), o.ctypes.data,
                         a.ctypes.data, t.ctypes.data)
    return st, o, a, t


@pytest.mark.parametrize("case", GEN_CASES)
def test_host_generator_equals_oracles(case):
    from oracle import generator_oracle as G
    B, H, W, A, density, seed = case
    o, a, t = generate_instances(B, H, W, A, density, seed)
    ro, ra, rt = G.generate_batch(seed, B, H, W, A, density, env_index_base=0, max_retries=50)
    assert np.array_equal(o, ro) and np.array_equal(a, ra) and np.array_equal(t, rt)
    st, co, ca, ct = _c_generate(B, H, W, A, density, seed)
    assert st == 0 and np.array_equal(o, co) and np.array_equal(a, ca) and np.array_equal(t, ct)


def test_generator_oracle_epochs_and_given_map():
    from oracle import generator_oracle as G
    m = np.zeros((5, 7), np.uint8)
    m[2, :] = 1
    for epoch in (0, 1, 5):
        _, pa, pt = G.generate_instance(42, 0, 5, 7, 3, 0.0, epoch=epoch, given_map=m)
        st, _, ca, ct = _c_generate(1, 5, 7, 3, 0.0, 42, given_map=m, epochs=[epoch])
        assert st == 0 and np.array_equal(pa, ca[0]) and np.array_equal(pt, ct[0])
    a0 = G.generate_instance(42, 0, 5, 7, 3, 0.0, epoch=0, given_map=m)[1]
    a1 = G.generate_instance(42, 0, 5, 7, 3, 0.0, epoch=1, given_map=m)[1]
    assert not np.array_equal(a0, a1), "a new generation draws a new placement"
    with pytest.raises(OverflowError):
        G.generate_instance(0, 0, 4, 4, 9, 0.0)
    with pytest.raises(OverflowError):
        G.generate_instance(0, 0, 4, 4, 4, 0.9, max_retries=3)


def test_generator_labels_are_min_indices():
    from oracle import generator_oracle as G
    rng = np.random.default_rng(0)
    obst = (rng.random((12, 15)) < 0.4).astype(np.uint8)
    lab = G.min_index_labels(obst)
    ref, pts = label_components(obst)
    for comp in pts:
        idx = [x * 15 + y for x, y in comp]
        assert all(lab[x, y] == min(idx) for x, y in comp)
    assert (lab[obst == 1] == -1).all()


def test_possible_positions_host_equals_oracle():
    """`possible_agents_xy` / `possible_targets_xy` (host numpy path of the product) == its oracle restatement."""
    from oracle import generator_oracle as G
    from pogema_amd.generator_host import place_from_possible
    pa = [(0, 0), (0, 3), (2, 2), (4, 1), (5, 5), (1, 4)]
    pt = [(3, 3), (0, 1), (5, 0), (2, 4)]
    agents, targets = place_from_possible(6, 100, pa, pt, 3)
    for b in range(6):
        ra, rt = G.place_from_possible(100, b, pa, pt, 3)
        assert np.array_equal(agents[b], ra) and np.array_equal(targets[b], rt)
        assert len({tuple(p) for p in agents[b]}) == 3 and len({tuple(p) for p in targets[b]}) == 3
        assert all(tuple(p) in pa for p in agents[b].tolist()) and all(tuple(p) in pt for p in targets[b].tolist())
    full, _ = place_from_possible(1, 5, pa, pt, 4)  # all four targets are used when A == len(list)
    with pytest.raises(OverflowError):
        place_from_possible(1, 0, pa, pt, 5)


def test_host_generator_fuzz_against_oracles():
    """Random geometries / densities / seeds: host generator == plain-C port == (for the small ones) the literal Python
    statement of GEN v2, including instances that need several attempts."""
    from oracle import generator_oracle as G
    rng = np.random.default_rng(2024)
    checked_py = 0
    for _ in range(60):
        H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        density = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7]))
        free = int(H * W * (1 - density))
        if free < 4:
            continue
        A = int(rng.integers(1, max(2, min(200, free // 3))))
        B = int(rng.integers(1, 6))
        seed = int(rng.integers(0, 2 ** 40))
        o = np.empty((B, H, W), np.uint8)
        a = np.empty((B, A, 2), np.int32)
        t = np.empty((B, A, 2), np.int32)
        lib = _lib.load()
        st = lib.pgx_generate(B, H, W, A, C.c_float(density), seed, 0, 6, 2, o.ctypes.data, a.ctypes.data, t.ctypes.data)
        cst, co, ca, ct = _c_generate(B, H, W, A, density, seed, max_retries=6)
        assert (st == 0) == (cst == 0), (H, W, A, density, seed)
        if st != 0:
            continue
        assert np.array_equal(o, co) and np.array_equal(a, ca) and np.array_equal(t, ct), (H, W, A, density, seed)
        if H * W <= 300 and checked_py < 15:
            ro, ra, rt = G.generate_batch(seed, B, H, W, A, density, env_index_base=0, max_retries=6)
            assert np.array_equal(o, ro) and np.array_equal(a, ra) and np.array_equal(t, rt)
            checked_py += 1
    assert checked_py >= 5
