"""ctypes binding of libpogema_amd.so (the C-ABI declared in include/pogema_amd.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be loaded the import of
the engine raises, and every compute call raises `PgxError` when the device is unusable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGX_LIB: diagnostic override (A/B of two builds of the SAME engine on one box); never a fallback
LIB_PATH = os.environ.get("PGX_LIB") or os.path.join(_HERE, "libpogema_amd.so")

PGX_ABI_VERSION = 6

COLLISION_SYSTEMS = {"priority": 0, "block_both": 1, "soft": 2}
ON_TARGET = {"finish": 0, "restart": 1, "nothing": 2}
ACTION_DTYPES = {"int8": 0, "int32": 1, "int64": 2}
SOFT_VERTEX_RULES = {"lowest_index": 0, "all_stay": 1}
SOFT_OCCUPANCY = {"index_order": 0, "exact": 1}
COOP_REWARDS = {"all_solved": 0, "per_agent": 1}
BAD_ACTIONS = {"noop": 0, "flag": 1}
LIFELONG_RNGS = {"build": 0, "numpy": 1}
def _obs_dtypes():
    import torch
    return {torch.float32: 0, torch.uint8: 1, torch.bfloat16: 2, torch.float16: 3}


def obs_elem_bytes(dtype) -> int:
    """Bytes per observation cell for a torch dtype the engine writes (float32 4, bfloat16 / float16 2, uint8 1)."""
    import torch
    return {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.uint8: 1}[dtype]


class _LazyObsDtypes(dict):
    """torch dtype -> PGX_OBS_* code; built on first use so that importing this module stays cheap."""

    def _fill(self):
        if not dict.__len__(self):
            self.update(_obs_dtypes())

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)


OBS_DTYPES = _LazyObsDtypes()
METRIC_NAMES = ("ISR", "CSR", "ep_length", "SoC", "makespan", "avg_throughput")
# hard limits of the engine (include/pogema_amd.h: PGX_MAX_*); upstream's GridConfig admits size 2..1024, obs_radius 1..128,
# num_agents >= 1 -- README.md "Limits"
MAX_OBS_RADIUS, MAX_AGENTS, MAX_SIDE = 15, 1024, 1024

# every symbol include/pogema_amd.h declares; tests/test_abi.py checks the library exports them all
EXPORTED_SYMBOLS = (
    "pgx_abi_version", "pgx_last_error", "pgx_create", "pgx_check_config", "pgx_destroy", "pgx_obs_elems", "pgx_agent_elems",
    "pgx_reset_from_state", "pgx_reset_random", "pgx_regenerate", "pgx_regenerate_failures", "pgx_get_map", "pgx_step", "pgx_observe", "pgx_set_metrics_buffers", "pgx_get_state", "pgx_generate", "pgx_place_agents",
    "pgx_snapshot_bytes", "pgx_save_snapshot", "pgx_load_snapshot", "pgx_time_observe", "pgx_bad_action_count",
    "pgx_buffers_create", "pgx_buffers_ptr", "pgx_buffers_get_info", "pgx_buffers_destroy", "pgx_set_targets",
    "pgx_np_streams", "pgx_np_streams_host", "pgx_np_generate", "pgx_np_generate_host", "pgx_rollout", "pgx_buffers_stride", "pgx_buffers_drop", "pgx_xcd_shares", "pgx_xcd_tune", "pgx_buffers_create_at", "pgx_time_observe_pair", "pgx_buffers_va_reserved", "pgx_get_geometry",
)


class PgxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pogema_amd engine error {code}: {message}")
        self.code = code


class PgxBuffersInfo(C.Structure):
    _fields_ = [("bytes", C.c_int64), ("count", C.c_int32), ("spread", C.c_int32), ("candidates", C.c_int32),
                ("reserved0", C.c_int32), ("same_zone_us", C.c_float), ("final_us", C.c_float), ("spacer_gib", C.c_double),
                ("buffer_gbs", C.c_float), ("reserved1", C.c_float)]


class PgxGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("lanes_per_env", "waves", "envs_per_wave", "multi_wave", "p16", "stagger",
                                          "store_policy", "state_stores", "grid", "lds_bytes", "for_rollout", "xcd_aware")]


class PgxRolloutIO(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("obs", C.c_void_p), ("rewards", C.c_void_p), ("terminated", C.c_void_p),
                ("truncated", C.c_void_p), ("is_active", C.c_void_p), ("episode_done", C.c_void_p), ("metrics", C.c_void_p),
                ("action_dtype", C.c_int32), ("obs_slots", C.c_int32), ("obs_slot_stride", C.c_int64),
                ("policy_seed", C.c_uint64), ("policy_step0", C.c_int64), ("actions_out", C.c_void_p)]


class PgxConfig(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("num_agents", C.c_int32),
        ("obs_radius", C.c_int32), ("collision_system", C.c_int32), ("on_target", C.c_int32),
        ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32), ("obs_dtype", C.c_int32),
        ("seed", C.c_uint64), ("env_index_base", C.c_int64),
        ("random_outside", C.c_int32), ("outside_density", C.c_float),
        ("soft_vertex_rule", C.c_int32), ("coop_reward", C.c_int32), ("bad_action", C.c_int32), ("lifelong_rng", C.c_int32),
        ("soft_occupancy", C.c_int32), ("abi_version", C.c_int32),
    ]


_lib = None


def build_hint() -> str:
    return ("build it with `python -c 'import __graft_entry__ as g; g.build()'` from the repo root "
            "(or `make -C pogema_amd/csrc`)")


def load() -> C.CDLL:
    """Load the engine library once; raises ImportError loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP engine is not built; {build_hint()}")
    # torch ships its own libamdhip64.so.7; load it FIRST so that the dynamic loader resolves our
    # NEEDED libamdhip64.so.7 to the very same runtime (two HIP runtimes in one process cannot both
    # own the GPU: the second reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    lib.pgx_abi_version.restype = C.c_int
    lib.pgx_last_error.restype = C.c_char_p
    lib.pgx_create.argtypes = [C.POINTER(PgxConfig), C.c_int, C.POINTER(vp)]
    lib.pgx_destroy.argtypes = [vp]
    lib.pgx_check_config.argtypes = [C.POINTER(PgxConfig)]
    lib.pgx_obs_elems.argtypes = [vp]
    lib.pgx_obs_elems.restype = i64
    lib.pgx_agent_elems.argtypes = [vp]
    lib.pgx_agent_elems.restype = i64
    lib.pgx_reset_from_state.argtypes = [vp, vp, vp, vp, vp]
    lib.pgx_reset_random.argtypes = [vp, f32, u64, vp, vp, i32, vp]
    lib.pgx_regenerate.argtypes = [vp, vp, f32, u64, vp, i32, vp, vp]
    lib.pgx_regenerate_failures.argtypes = [vp, vp]
    lib.pgx_regenerate_failures.restype = i64
    lib.pgx_get_map.argtypes = [vp, vp, vp]
    lib.pgx_buffers_create.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_double, C.POINTER(vp)]
    lib.pgx_buffers_create.restype = C.c_int
    lib.pgx_buffers_create_at.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double, C.POINTER(vp)]
    lib.pgx_buffers_create_at.restype = C.c_int
    lib.pgx_buffers_ptr.argtypes = [vp, C.c_int]
    lib.pgx_buffers_ptr.restype = vp
    lib.pgx_buffers_get_info.argtypes = [vp, C.POINTER(PgxBuffersInfo)]
    lib.pgx_buffers_get_info.restype = C.c_int
    lib.pgx_buffers_destroy.argtypes = [vp]
    lib.pgx_buffers_destroy.restype = C.c_int
    lib.pgx_np_streams.argtypes = [vp, i64, i32, u64, C.c_double, i64, vp, vp]
    lib.pgx_np_streams.restype = C.c_int
    lib.pgx_np_streams_host.argtypes = [vp, i64, i32, u64, C.c_double, i64, vp]
    lib.pgx_np_streams_host.restype = C.c_int
    lib.pgx_np_generate.argtypes = [vp, i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp, vp, vp]
    lib.pgx_np_generate.restype = C.c_int
    lib.pgx_np_generate_host.argtypes = [vp, i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp, vp]
    lib.pgx_np_generate_host.restype = C.c_int
    lib.pgx_xcd_tune.argtypes = [vp, vp, vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]
    lib.pgx_xcd_tune.restype = C.c_int
    lib.pgx_get_geometry.argtypes = [vp, i32, C.POINTER(PgxGeometry)]
    lib.pgx_get_geometry.restype = C.c_int
    lib.pgx_xcd_shares.argtypes = [vp, vp]
    lib.pgx_xcd_shares.restype = C.c_int
    lib.pgx_buffers_drop.argtypes = [vp, i32]
    lib.pgx_buffers_drop.restype = C.c_int
    lib.pgx_buffers_va_reserved.argtypes = []
    lib.pgx_buffers_va_reserved.restype = C.c_int64
    lib.pgx_buffers_stride.argtypes = [vp]
    lib.pgx_buffers_stride.restype = C.c_int64
    lib.pgx_rollout.argtypes = [vp, i32, C.POINTER(PgxRolloutIO), vp]
    lib.pgx_rollout.restype = C.c_int
    lib.pgx_set_targets.argtypes = [vp, vp, vp, vp]
    lib.pgx_set_targets.restype = C.c_int
    lib.pgx_bad_action_count.argtypes = [vp, vp]
    lib.pgx_bad_action_count.restype = i64
    lib.pgx_time_observe.argtypes = [vp, vp, i32, C.POINTER(C.c_float), vp]
    lib.pgx_time_observe.restype = C.c_int
    lib.pgx_time_observe_pair.argtypes = [vp, vp, vp, i32, C.POINTER(C.c_float), vp]
    lib.pgx_time_observe_pair.restype = C.c_int
    lib.pgx_snapshot_bytes.argtypes = [vp]
    lib.pgx_snapshot_bytes.restype = i64
    lib.pgx_save_snapshot.argtypes = [vp, vp, vp]
    lib.pgx_load_snapshot.argtypes = [vp, vp, vp]
    lib.pgx_save_snapshot.restype = lib.pgx_load_snapshot.restype = C.c_int
    lib.pgx_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.pgx_observe.argtypes = [vp, vp, vp]
    lib.pgx_set_metrics_buffers.argtypes = [vp, vp, vp]
    lib.pgx_get_state.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.pgx_generate.argtypes = [i32, i32, i32, i32, f32, u64, i64, i32, i32, vp, vp, vp]
    lib.pgx_place_agents.argtypes = [i32, i32, i32, i32, u64, i64, i32, i32, vp, i32, vp, vp]
    for name in ("pgx_create", "pgx_destroy", "pgx_reset_from_state", "pgx_reset_random", "pgx_regenerate", "pgx_get_map",
                 "pgx_step", "pgx_observe", "pgx_set_metrics_buffers", "pgx_set_metrics_buffers",
                 "pgx_get_state", "pgx_generate", "pgx_place_agents"):
        getattr(lib, name).restype = C.c_int
    if lib.pgx_abi_version() != PGX_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.pgx_abi_version()} != expected {PGX_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        msg = load().pgx_last_error()
        raise PgxError(status, msg.decode("utf-8", "replace") if msg else "unknown error")
