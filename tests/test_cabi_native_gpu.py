"""GPU: a native host program (tests/native/cabi_roundtrip.cpp -- no Python, no torch, only include/pogema_amd.h and
hipMalloc'd pointers) drives reset + steps through the C-ABI and prints checksums; the same scenario is replayed here
on the CPU oracle (generator GEN v2 + C oracle rollout) and the checksums must match bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from util import c_oracle_rollout

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "native", "cabi_roundtrip")
COLL = ("priority", "block_both", "soft")
ONT = ("finish", "restart", "nothing")


def _fnv(h, arr):
    data = np.ascontiguousarray(arr).tobytes()
    for byte in data:  # small scenarios only
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _generate(B, S, A, density, seed, base):
    from oracle.c_oracle import load
    lib = load()
    lib.po_generate.argtypes = [C.c_int32] * 4 + [C.c_float, C.c_uint64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.po_generate.restype = C.c_int
    o = np.empty((B, S, S), np.uint8)
    a = np.empty((B, A, 2), np.int32)
    t = np.empty((B, A, 2), np.int32)
    assert lib.po_generate(B, S, S, A, density, seed, base, None, 10, 0, o.ctypes.data, a.ctypes.data, t.ctypes.data) == 0
    return o, a, t


@pytest.mark.parametrize("case", [(6, 12, 5, 3, 2, 0, 12, 41), (3, 20, 70, 2, 0, 1, 9, 7), (4, 9, 3, 4, 1, 2, 10, 3)],
                         ids=["soft_finish", "priority_restart_2waves", "block_both_nothing"])
def test_native_program_matches_oracle(case):
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.dirname(EXE)], check=True)
    B, S, A, r, coll, ont, T, seed = case
    out = subprocess.run([EXE] + [str(v) for v in case], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = dict(kv.split("=") for kv in out.stdout.split())
    base = 5  # cfg.env_index_base in the program
    o, a, t = _generate(B, S, A, 0.3, seed, base)
    gi = np.arange(B * A)
    actions = np.stack([((tt * 7 + (gi // A) * 3 + (gi % A) * 5 + ((tt + gi) >> 2)) % 5).reshape(B, A)
                        for tt in range(T)]).astype(np.int64)
    ref = c_oracle_rollout(o, a, t, actions, obs_radius=r, collision_system=COLL[coll], on_target=ONT[ont],
                           max_episode_steps=7, auto_reset=True, seed=seed, env_index_base=base)
    FNV0 = 0xCBF29CE484222325
    h_reset = _fnv(_fnv(_fnv(FNV0, o), a), t)
    h_obs = _fnv(FNV0, ref["obs0"].astype(np.float32))
    h_flags, h_state = FNV0, FNV0
    for tt in range(T):
        h_obs = _fnv(h_obs, ref["obs"][tt].astype(np.float32))
        h_flags = _fnv(h_flags, ref["rewards"][tt].astype(np.float32))
        for k in ("terminated", "truncated", "is_active"):
            h_flags = _fnv(h_flags, ref[k][tt].astype(np.uint8))
        h_state = _fnv(h_state, ref["agents_xy"][tt].astype(np.int32))
        h_state = _fnv(h_state, ref["targets_xy"][tt].astype(np.int32))
        h_state = _fnv(h_state, ref["elapsed"][tt].astype(np.int32))
    assert got["reset"] == f"{h_reset:016x}", "generator / placement"
    assert got["state"] == f"{h_state:016x}", "agent cells, targets, step counters"
    assert got["flags"] == f"{h_flags:016x}", "rewards / terminated / truncated / is_active"
    assert got["obs"] == f"{h_obs:016x}", "observations"
