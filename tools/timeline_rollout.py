"""Diagnostic: where one iteration of a pgx_rollout launch spends its time.  The per-workgroup stamp
slots (PGX_FLAGS bit 2; bit 6 = state-phase stamps) are overwritten by every iteration of the on-device loop, so what
is read back is the LAST iteration of every workgroup; the iteration period is the launch time / K.
usage: python tools/timeline_rollout.py cfg1 [cfg3 ...]     (K=64 by default, env K)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema, _lib
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
K = int(os.environ.get("K", "64"))
lib = _lib.load()
lib.pgx_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
base_flags = int(os.environ.get("PGX_FLAGS", "0"), 0)
for name in sys.argv[1:] or ["cfg1", "cfg3"]:
    batch, size, agents, r = WL[name]
    for flags, labels in ((4, ("iteration starts", "state phase done", "row masks done / stream starts", "own stores acknowledged")),
                          (4 | 64, ("iteration starts", "loads+staging done", "collisions resolved", "state phase done"))):
        os.environ["PGX_FLAGS"] = str(base_flags | flags)
        env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft",
                                   max_episode_steps=64), batch=batch, auto_reset=True, reuse_buffers=True)
        env.reset(seed=0)
        acts = torch.randint(0, 5, (K, batch, agents), device="cuda", dtype=torch.int8)
        env.rollout(acts[:4], obs_slots=2)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = env.rollout(acts, obs_slots=2)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / K)
            del out
        buf = np.zeros((batch, 4), dtype=np.uint64)
        _lib.check(lib.pgx_debug_timestamps(env._handle, buf.ctypes.data, buf.size))
        buf = buf[buf[:, 0] != 0].astype(np.int64)
        d = (buf - buf[:, :1]) / 100.0  # us since the workgroup's own iteration start (wall_clock64: 100 MHz)
        print(f"{name} flags={flags}: {len(buf)} workgroups, rollout {best:.2f} us per step (stamps on), last iteration of each workgroup:")
        for lab, col in zip(labels, range(4)):
            q = np.percentile(d[:, col], [10, 50, 90, 100])
            print(f"  {lab:32s} since iteration start p10/p50/p90/max = " + " / ".join(f"{v:7.2f}" for v in q))
        env.close()
