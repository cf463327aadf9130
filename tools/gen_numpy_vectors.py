#!/usr/bin/env python3
"""Writes tests/golden/numpy_rng_vectors.npz: outputs of numpy's own Generator (np.random.default_rng(seed)) for the
primitives the engine re-implements (pogema_amd/csrc/pgx_nprng.h).  numpy IS the reference for this layer (upstream
POGEMA draws from it), it is installed in the build image, and the vectors are data: inputs (seeds, op, parameters) and
numpy's outputs -- nothing of numpy's code.  tests/test_nprng.py (host) and tests/test_nprng_gpu.py (device) compare
bit for bit.  Re-run after a numpy upgrade; the numpy version is recorded in the file."""
import os

import numpy as np

SEEDS = np.array([0, 1, 2, 42, 12345, 2 ** 31 - 1, 2 ** 32, 2 ** 40 + 5, 2 ** 63 + 11, 2 ** 64 - 1], dtype=np.uint64)
DRAWS = 257
CASES = [  # name, op, n, p
    ("uint64", "uint64", 1, 0.0), ("random", "random", 1, 0.0),
    ("integers_5", "integers", 5, 0.0), ("integers_37", "integers", 37, 0.0), ("integers_4096", "integers", 4096, 0.0),
    ("integers_2p32", "integers", 2 ** 32, 0.0), ("integers_3e9", "integers", 3_000_000_000, 0.0),
    ("integers_2p40", "integers", 2 ** 40 + 7, 0.0), ("integers_1", "integers", 1, 0.0),
    ("binomial_0.3", "binomial1", 1, 0.3), ("binomial_0.05", "binomial1", 1, 0.05), ("binomial_0.5", "binomial1", 1, 0.5),
    ("binomial_0.8", "binomial1", 1, 0.8), ("binomial_1.0", "binomial1", 1, 1.0), ("binomial_0.0", "binomial1", 1, 0.0),
    ("permutation", "permutation", 1, 0.0),
]


def numpy_reference(seed, op, n, p, draws):
    g = np.random.default_rng(int(seed))
    if op == "uint64":
        return g.bit_generator.random_raw(draws).astype(np.uint64)
    if op == "random":
        return g.random(draws)
    if op == "integers":
        return g.integers(0, n, size=draws)
    if op == "binomial1":
        return g.binomial(1, p, size=draws)
    if op == "permutation":
        return g.permutation(draws)
    raise ValueError(op)


def main():
    out = {"numpy_version": np.array(np.__version__), "seeds": SEEDS, "draws": np.array(DRAWS),
           "case_names": np.array([c[0] for c in CASES]), "case_ops": np.array([c[1] for c in CASES]),
           "case_n": np.array([c[2] for c in CASES], dtype=np.uint64), "case_p": np.array([c[3] for c in CASES])}
    for name, op, n, p in CASES:
        out["out_" + name] = np.stack([numpy_reference(s, op, n, p, DRAWS) for s in SEEDS])
    # SeedSequence / PCG64 seeding on their own (first state words), for a sharper failure message than "draw 0 differs"
    out["pool"] = np.stack([np.random.SeedSequence(int(s)).pool for s in SEEDS]).astype(np.uint32)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "numpy_rng_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; numpy", np.__version__)


if __name__ == "__main__":
    main()
