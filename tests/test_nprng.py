"""CPU: the engine's numpy-compatible random primitives (pgx_np_streams_host; pogema_amd/csrc/pgx_nprng.h) against
numpy ITSELF -- the committed vectors of tools/gen_numpy_vectors.py and live draws.  Bit-exact."""
import os

import numpy as np
import pytest

from pogema_amd.nprng import np_streams_host

VEC = os.path.join(os.path.dirname(__file__), "golden", "numpy_rng_vectors.npz")


def _cases():
    z = np.load(VEC, allow_pickle=False)
    return [(str(n), str(o), int(k), float(p)) for n, o, k, p in zip(z["case_names"], z["case_ops"], z["case_n"], z["case_p"])]


@pytest.mark.parametrize("name,op,n,p", _cases(), ids=[c[0] for c in _cases()])
def test_host_primitives_equal_committed_numpy_vectors(name, op, n, p):
    z = np.load(VEC, allow_pickle=False)
    got = np_streams_host(z["seeds"], op, int(z["draws"]), n=n, p=p)
    ref = z["out_" + name]
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint64), ref.astype(got.dtype).view(np.uint64)), f"{name}: differs from numpy {z['numpy_version']}"


def test_vectors_match_the_installed_numpy():
    """The fixture is numpy's output: regenerate a case live and compare (catches a stale file after a numpy upgrade)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_numpy_vectors import numpy_reference
    z = np.load(VEC, allow_pickle=False)
    for name, op, n, p in _cases():
        live = np.stack([numpy_reference(s, op, n, p, int(z["draws"])) for s in z["seeds"][:3]])
        assert np.array_equal(live, z["out_" + name][:3]), name


def test_many_live_draws_and_call_shapes():
    """>= 10^5 draws across random seeds against live numpy; scalar calls, choice(), list shuffle and array shuffle
    consume the stream exactly like the array forms."""
    rng = np.random.default_rng(2024)
    seeds = rng.integers(0, 2 ** 63, size=40, dtype=np.uint64)
    draws = 3000  # 40 x 3000 = 1.2e5 per op
    for op, kw, ref in (("integers", dict(n=97), lambda g: g.integers(0, 97, size=draws)),
                        ("integers", dict(n=2 ** 33 + 1), lambda g: g.integers(0, 2 ** 33 + 1, size=draws)),
                        ("random", {}, lambda g: g.random(draws)),
                        ("binomial1", dict(p=0.3), lambda g: g.binomial(1, 0.3, size=draws)),
                        ("permutation", {}, lambda g: g.permutation(draws))):
        got = np_streams_host(seeds, op, draws, **kw)
        for i, s in enumerate(seeds):
            assert np.array_equal(got[i], ref(np.random.default_rng(int(s))).astype(got.dtype)), (op, kw, int(s))
    g = np.random.default_rng(7)
    assert [int(g.integers(0, 37)) for _ in range(64)] == np_streams_host([7], "integers", 64, n=37)[0].tolist()
    g = np.random.default_rng(7)
    assert [int(g.choice(37)) for _ in range(64)] == np_streams_host([7], "integers", 64, n=37)[0].tolist()
    cells = [(i // 7, i % 7) for i in range(40)]
    g = np.random.default_rng(9)
    picked = [tuple(int(v) for v in g.choice(cells, 1)[0]) for _ in range(20)]  # upstream's `rng.choice(component, 1)`
    assert picked == [cells[k] for k in np_streams_host([9], "integers", 20, n=len(cells))[0]]
    x = list(range(50))
    np.random.default_rng(11).shuffle(x)
    assert x == np_streams_host([11], "permutation", 50)[0].tolist()
    y = np.arange(50)
    np.random.default_rng(11).shuffle(y)
    assert y.tolist() == x


def test_bad_arguments():
    from pogema_amd._lib import PgxError
    with pytest.raises(PgxError):
        np_streams_host([1], "integers", 4, n=0)
    with pytest.raises(PgxError):
        np_streams_host([1], "binomial1", 4, p=1.5)


from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=150, deadline=None)
@given(seed=st.integers(0, 2 ** 64 - 1), n=st.one_of(st.integers(1, 2 ** 16), st.integers(1, 2 ** 63 - 1), st.sampled_from([2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1])),
       p=st.floats(0.0, 1.0), draws=st.integers(1, 64))
def test_primitives_property(seed, n, p, draws):
    """Arbitrary seeds / bounds / probabilities: every primitive equals numpy's Generator draw for draw."""
    assert np.array_equal(np_streams_host([seed], "integers", draws, n=n)[0],
                          np.random.default_rng(seed).integers(0, n, size=draws))
    assert np.array_equal(np_streams_host([seed], "binomial1", draws, p=p)[0],
                          np.random.default_rng(seed).binomial(1, p, size=draws))
    assert np.array_equal(np_streams_host([seed], "permutation", draws)[0], np.random.default_rng(seed).permutation(draws))
    assert np.array_equal(np_streams_host([seed], "random", draws)[0], np.random.default_rng(seed).random(draws))
