"""numpy-compatible random primitives of the engine (pgx_np_streams*, pogema_amd/csrc/pgx_nprng.h): many independent
`np.random.default_rng(seed)` streams advanced in parallel, bit-identical with numpy (tests/test_nprng*.py).

    np_streams_host(seeds, "integers", n=10, draws=5)[s]  ==  np.random.default_rng(seeds[s]).integers(0, 10, size=5)
"""
from __future__ import annotations

import numpy as np

from . import _lib

OPS = {"uint64": 0, "random": 1, "integers": 2, "binomial1": 3, "permutation": 4}
_OUT = {"uint64": np.uint64, "random": np.float64, "integers": np.int64, "binomial1": np.int64, "permutation": np.int64}


def np_streams_host(seeds, op: str, draws: int, n: int = 1, p: float = 0.0) -> np.ndarray:
    """Host evaluation (same arithmetic as the device kernel); returns [len(seeds), draws]."""
    lib = _lib.load()
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    out = np.empty((len(seeds), draws), dtype=_OUT[op])
    _lib.check(lib.pgx_np_streams_host(seeds.ctypes.data, len(seeds), OPS[op], int(n), float(p), int(draws), out.ctypes.data))
    return out


def np_streams(seeds, op: str, draws: int, n: int = 1, p: float = 0.0, device="cuda:0"):
    """Device evaluation, one GPU thread per stream; `seeds` any integer sequence / tensor; returns a torch tensor."""
    import ctypes as C
    import torch
    lib = _lib.load()
    dev = torch.device(device)
    s = torch.as_tensor(np.ascontiguousarray(seeds, dtype=np.uint64).view(np.int64)).to(dev)
    dtype = {"uint64": torch.int64, "random": torch.float64}.get(op, torch.int64)  # uint64 bits are returned in an int64 tensor
    out = torch.empty((s.numel(), draws), dtype=dtype, device=dev)
    _lib.check(lib.pgx_np_streams(s.data_ptr(), s.numel(), OPS[op], int(n), float(p), int(draws), out.data_ptr(),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
