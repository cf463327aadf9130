"""Diagnostic: where the state phase of pgx::step_kernel spends its time (PGX_FLAGS bits 2 + 6: per-workgroup stamps
after the loads/staging, after the collision resolve, at the end of the state phase), observation write off/on."""
import ctypes as C, os, sys
import numpy as np
os.environ["PGX_FLAGS"] = str(int(os.environ.get("PGX_FLAGS", "0"), 0) | 4 | 64)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema, _lib
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5)}
batch, size, agents, r = WL[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                batch=batch, auto_reset=True, reuse_buffers=True)
env.reset(seed=0)
acts = torch.randint(0, 5, (batch, agents), device="cuda")
lib = _lib.load()
lib.pgx_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
for with_obs in (False, True):
    for _ in range(5):
        env.step(acts, compute_obs=with_obs)
    torch.cuda.synchronize()
    buf = np.zeros((batch, 4), dtype=np.uint64)
    _lib.check(lib.pgx_debug_timestamps(env._handle, buf.ctypes.data, buf.size))
    buf = buf[buf[:, 0] != 0]
    t = (buf.astype(np.int64) - int(buf[:, 0].min())) / 100.0
    print(f"observation write {'ON' if with_obs else 'OFF'}: {len(buf)} workgroups")
    for name, col in (("start", 0), ("loads+staging done", 1), ("collisions resolved", 2), ("state phase done", 3)):
        q = np.percentile(t[:, col], [0, 10, 50, 90, 100])
        print(f"  {name:20s} min/p10/p50/p90/max = " + " / ".join(f"{v:7.1f}" for v in q))
