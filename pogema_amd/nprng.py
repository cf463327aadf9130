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


def np_generate_host(seeds, height: int, width: int, num_agents: int, density: float, given_map=None):
    """Instances the way upstream draws them (recalled; pgx_np_generate_host): returns (obstacles u8 [B,H,W],
    agents_xy i32 [B,A,2], targets_xy i32 [B,A,2], status i32 [B]; status 1 = not enough start/target pairs)."""
    lib = _lib.load()
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    B = len(seeds)
    obstacles = np.zeros((B, height, width), np.uint8)
    agents = np.zeros((B, num_agents, 2), np.int32)
    targets = np.zeros((B, num_agents, 2), np.int32)
    scratch = np.empty(4 * height * width, np.uint32)
    status = np.zeros(B, np.int32)
    gm = None
    if given_map is not None:
        gm = np.ascontiguousarray(np.asarray(given_map) != 0, dtype=np.uint8)
        if gm.shape != (height, width):
            raise ValueError(f"given_map must be [{height}, {width}]")
    _lib.check(lib.pgx_np_generate_host(seeds.ctypes.data, B, height, width, num_agents, float(density),
                                        gm.ctypes.data if gm is not None else None, obstacles.ctypes.data,
                                        agents.ctypes.data, targets.ctypes.data, scratch.ctypes.data, status.ctypes.data))
    return obstacles, agents, targets, status


def np_generate(seeds, height: int, width: int, num_agents: int, density: float, device="cuda:0", given_map=None):
    """The same on the device, one GPU thread per env; returns torch tensors."""
    import ctypes as C
    import torch
    lib = _lib.load()
    dev = torch.device(device)
    s = torch.as_tensor(np.ascontiguousarray(seeds, dtype=np.uint64).view(np.int64)).to(dev)
    B = s.numel()
    obstacles = torch.zeros((B, height, width), dtype=torch.uint8, device=dev)
    agents = torch.zeros((B, num_agents, 2), dtype=torch.int32, device=dev)
    targets = torch.zeros((B, num_agents, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty((B, 4 * height * width), dtype=torch.int32, device=dev)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    gm = None
    if given_map is not None:
        gm = torch.as_tensor(np.asarray(given_map) != 0).to(dev).to(torch.uint8).contiguous()
        if tuple(gm.shape) != (height, width):
            raise ValueError(f"given_map must be [{height}, {width}]")
    _lib.check(lib.pgx_np_generate(s.data_ptr(), B, height, width, num_agents, float(density),
                                   gm.data_ptr() if gm is not None else None, obstacles.data_ptr(),
                                   agents.data_ptr(), targets.data_ptr(), scratch.data_ptr(), status.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return obstacles, agents, targets, status
