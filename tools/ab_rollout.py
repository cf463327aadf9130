"""Diagnostic: in-process, interleaved A/B of pgx_rollout between engine variants (PGX_* knobs read at pgx_create, or
another BUILD of the library: "PGX_LIB=pogema_amd/libpogema_amd_r5.so"), all variants writing into the SAME observation
ring so that buffer placement cancels (as tools/ab_inproc.py does for pgx_step).
usage: python tools/ab_rollout.py cfg3 "" "PGX_LIB=pogema_amd/libpogema_amd_r5.so"      (env: K=64 SLOTS=2 ROUNDS=8)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pogema_amd import GridConfig, VecPogema  # noqa: E402
from pogema_amd import _lib as _L  # noqa: E402

WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7),
      "a8big": (65536, 16, 8, 5), "a16small": (1024, 32, 16, 5), "a32mid": (2048, 32, 32, 5), "a4": (4096, 12, 4, 3)}
K, SLOTS, ROUNDS = int(os.environ.get("K", "64")), int(os.environ.get("SLOTS", "2")), int(os.environ.get("ROUNDS", "8"))
wl, variants = sys.argv[1], sys.argv[2:]
batch, size, agents, r = WL[wl]
DEFAULT_LIB = _L.LIB_PATH
KNOBS = ("PGX_ROLL_PC", "PGX_BIG", "PGX_FLAGS", "PGX_EPW", "PGX_STAGGER", "PGX_WAVES", "PGX_STORE", "PGX_STATE_STORES", "PGX_ROLL_RESIDENT")
envs = []
for v in variants:
    for k in KNOBS:
        os.environ.pop(k, None)
    lib_path = None
    for kv in v.split(","):
        if kv:
            k, val = kv.split("=")
            if k == "PGX_LIB":
                lib_path = os.path.abspath(val)
            else:
                os.environ[k] = val
    want = lib_path or DEFAULT_LIB
    if _L.LIB_PATH != want or _L._lib is None:
        _L._lib, _L.LIB_PATH = None, want  # VecPogema keeps the library it was created with
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft",
                               max_episode_steps=64), batch=batch, auto_reset=True, reuse_buffers=True)
    env.reset(seed=0)
    if envs:
        env._rollout_pools = envs[0]._rollout_pools  # the same ring for every variant
    envs.append(env)
acts = torch.randint(0, 5, (K, batch, agents), device="cuda", dtype=torch.int8)
for env in envs:
    env.rollout(acts[:4], obs_slots=SLOTS)
torch.cuda.synchronize()
print("ring:", envs[0].placement, flush=True)
times = np.zeros((len(envs), ROUNDS))
for rd in range(ROUNDS):
    for i, env in enumerate(envs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = env.rollout(acts, obs_slots=SLOTS)
        b.record()
        torch.cuda.synchronize()
        times[i, rd] = a.elapsed_time(b) * 1e3 / K
        del out
obs_bytes = batch * agents * 3 * (2 * r + 1) ** 2 * 4
for v, t in zip(variants, times):
    print(f"{wl} K={K} slots={SLOTS} {v or '(this build)':44s} median {np.median(t):8.2f} us/step  min {t.min():8.2f}  max {t.max():8.2f}"
          f"   obs stream alone at 8 TB/s: {obs_bytes / 8e12 * 1e6:.1f} us")
