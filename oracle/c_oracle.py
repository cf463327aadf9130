"""ctypes wrapper of oracle/libpogema_oracle.so (plain-C port of pogema_oracle.py).
TEST INFRASTRUCTURE ONLY -- parity unpinned, see pogema_oracle.py / DESIGN.md."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpogema_oracle.so")

COLLISION = {"priority": 0, "block_both": 1, "soft": 2}
ON_TARGET = {"finish": 0, "restart": 1, "nothing": 2}
SOFT_VERTEX_RULE = {"lowest_index": 0, "all_stay": 1}
COOP_REWARD = {"all_solved": 0, "per_agent": 1}
BAD_ACTION = {"noop": 0, "flag": 1}
SOFT_OCCUPANCY = {"index_order": 0, "exact": 1}


class PoConfig(C.Structure):
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("num_agents", C.c_int32),
                ("obs_radius", C.c_int32), ("collision_system", C.c_int32), ("on_target", C.c_int32),
                ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32), ("reserved0", C.c_int32),
                ("seed", C.c_uint64), ("env_index_base", C.c_int64), ("random_outside", C.c_int32),
                ("outside_density", C.c_float), ("soft_vertex_rule", C.c_int32), ("coop_reward", C.c_int32),
                ("bad_action", C.c_int32), ("soft_occupancy", C.c_int32)]


_lib = None


def load(build_if_missing: bool = True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle`")
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.po_create.argtypes = [C.POINTER(PoConfig)]
    lib.po_create.restype = vp
    lib.po_destroy.argtypes = [vp]
    lib.po_reset.argtypes = [vp, vp, vp, vp]
    lib.po_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.po_observe.argtypes = [vp, vp]
    lib.po_bad_action_count.argtypes = [vp]
    lib.po_bad_action_count.restype = C.c_int64
    lib.po_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


class COracle:
    """Batched CPU oracle with the same surface as the engine (numpy in / numpy out)."""

    def __init__(self, batch, height, width, num_agents, obs_radius, collision_system="priority", on_target="finish",
                 max_episode_steps=64, auto_reset=False, seed=0, env_index_base=0, empty_outside=True,
                 outside_density=0.0, soft_vertex_rule="lowest_index", coop_reward="all_solved", bad_action="noop",
                 lifelong_rng="build", soft_occupancy="index_order"):
        if lifelong_rng != "build":
            raise NotImplementedError("the plain-C port has the build's lifelong stream only; use the Python oracle")
        self.lib = load()
        self.B, self.H, self.Wd, self.A, self.r = batch, height, width, num_agents, obs_radius
        self.W = 2 * obs_radius + 1
        cfg = PoConfig(batch, height, width, num_agents, obs_radius, COLLISION[collision_system], ON_TARGET[on_target],
                       max_episode_steps, int(auto_reset), 0, seed, env_index_base, 0 if empty_outside else 1,
                       float(outside_density), SOFT_VERTEX_RULE[soft_vertex_rule], COOP_REWARD[coop_reward],
                       BAD_ACTION[bad_action], SOFT_OCCUPANCY[soft_occupancy])
        self.h = self.lib.po_create(C.byref(cfg))
        if not self.h:
            raise MemoryError("po_create failed")

    def close(self):
        if self.h:
            self.lib.po_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, obstacles, agents_xy, targets_xy):
        o = np.ascontiguousarray(obstacles, dtype=np.uint8)
        a = np.ascontiguousarray(agents_xy, dtype=np.int32)
        t = np.ascontiguousarray(targets_xy, dtype=np.int32)
        assert o.shape == (self.B, self.H, self.Wd) and a.shape == (self.B, self.A, 2) and t.shape == a.shape
        self.lib.po_reset(self.h, o.ctypes.data, a.ctypes.data, t.ctypes.data)
        return self.observe()

    def observe(self):
        obs = np.empty((self.B, self.A, 3, self.W, self.W), np.float32)
        self.lib.po_observe(self.h, obs.ctypes.data)
        return obs

    def step(self, actions, nthreads=1, compute_obs=True, out=None):
        acts = np.ascontiguousarray(actions, dtype=np.int64)
        assert acts.shape == (self.B, self.A)
        if out is None:
            out = (np.empty((self.B, self.A, 3, self.W, self.W), np.float32) if compute_obs else None,
                   np.empty((self.B, self.A), np.float32), np.empty((self.B, self.A), np.uint8),
                   np.empty((self.B, self.A), np.uint8), np.empty((self.B, self.A), np.uint8))
        obs, rew, term, trunc, act = out
        self.metrics = np.zeros((self.B, 6), np.float32)
        self.episode_done = np.zeros((self.B,), np.uint8)
        self.lib.po_step(self.h, acts.ctypes.data, obs.ctypes.data if obs is not None else None, rew.ctypes.data,
                         term.ctypes.data, trunc.ctypes.data, act.ctypes.data, self.metrics.ctypes.data,
                         self.episode_done.ctypes.data, int(nthreads))
        return obs, rew, term.astype(bool), trunc.astype(bool), act.astype(bool)

    def bad_action_count(self):
        """Out-of-range actions of active agents since the last call (bad_action='flag' only)."""
        return int(self.lib.po_bad_action_count(self.h))

    def get_state(self, occupancy=False):
        a = np.empty((self.B, self.A, 2), np.int32)
        t = np.empty((self.B, self.A, 2), np.int32)
        act = np.empty((self.B, self.A), np.uint8)
        el = np.empty((self.B,), np.int32)
        occ = np.empty((self.B, self.H + 2 * self.r, self.Wd + 2 * self.r), np.uint8) if occupancy else None
        self.lib.po_get_state(self.h, a.ctypes.data, t.ctypes.data, act.ctypes.data, el.ctypes.data,
                              occ.ctypes.data if occupancy else None)
        st = {"agents_xy": a, "targets_xy": t, "is_active": act.astype(bool), "elapsed": el}
        if occupancy:
            st["occupancy"] = occ
        return st
