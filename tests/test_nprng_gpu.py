"""GPU: the numpy-compatible primitives on the device (pgx_np_streams: one thread per default_rng(seed) stream) against
the committed numpy vectors and against live numpy for >= 10^5 draws across seeds.  Bit-exact."""
import os

import numpy as np
import pytest

from test_nprng import VEC, _cases

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("name,op,n,p", _cases(), ids=[c[0] for c in _cases()])
def test_device_primitives_equal_committed_numpy_vectors(name, op, n, p):
    from pogema_amd.nprng import np_streams
    z = np.load(VEC, allow_pickle=False)
    got = np_streams(z["seeds"], op, int(z["draws"]), n=n, p=p)
    ref = z["out_" + name]
    ref_bits = (ref if ref.dtype == np.float64 else ref.astype(np.int64)).view(np.uint64)
    assert np.array_equal(_bits(got), ref_bits), f"{name}: device differs from numpy {z['numpy_version']}"


def test_device_equals_live_numpy_over_many_draws():
    from pogema_amd.nprng import np_streams, np_streams_host
    rng = np.random.default_rng(77)
    seeds = rng.integers(0, 2 ** 64 - 1, size=512, dtype=np.uint64, endpoint=True)
    draws = 400  # 512 x 400 = 2.0e5 draws per op
    total = 0
    for op, kw, ref in (("integers", dict(n=1000003), lambda g: g.integers(0, 1000003, size=draws)),
                        ("integers", dict(n=2 ** 35 + 9), lambda g: g.integers(0, 2 ** 35 + 9, size=draws)),
                        ("random", {}, lambda g: g.random(draws)),
                        ("binomial1", dict(p=0.3), lambda g: g.binomial(1, 0.3, size=draws)),
                        ("binomial1", dict(p=0.7), lambda g: g.binomial(1, 0.7, size=draws)),
                        ("permutation", {}, lambda g: g.permutation(draws)),
                        ("uint64", {}, lambda g: g.bit_generator.random_raw(draws))):
        got = np_streams(seeds, op, draws, **kw)
        host = np_streams_host(seeds, op, draws, **kw)
        assert np.array_equal(_bits(got), host.view(np.uint64)), f"{op}: device != host arithmetic"
        for i in range(0, len(seeds), 7):  # numpy itself for every 7th stream (the host path is checked against numpy on CPU)
            live = ref(np.random.default_rng(int(seeds[i])))
            live_bits = (live if live.dtype == np.float64 else live.astype(np.uint64 if op == "uint64" else np.int64)).view(np.uint64)
            assert np.array_equal(_bits(got[i]), live_bits), (op, kw, int(seeds[i]))
        total += got.numel()
    assert total >= 100000
