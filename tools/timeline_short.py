"""Diagnostic: per-workgroup timeline of one step launch for the short configurations (PGX_FLAGS bit 2 = stamps; bit 6 =
state-phase stamps instead of stream stamps).  Prints both stamp sets plus the launch-to-launch time."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema, _lib
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7), "big": (1024, 1024, 256, 5), "mid": (1024, 640, 256, 5)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
batch, size, agents, r = WL[name]
lib = _lib.load()
lib.pgx_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
base_flags = int(os.environ.get("PGX_FLAGS", "0"), 0)
for flags, labels in ((4, ("start", "state phase done", "row masks done / stream starts", "own stores acknowledged")),
                      (4 | 64, ("start", "loads+staging done", "collisions resolved", "state phase done"))):
    os.environ["PGX_FLAGS"] = str(base_flags | flags)
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                    batch=batch, auto_reset=True, reuse_buffers=True)
    env.reset(seed=0)
    acts = torch.randint(0, 5, (batch, agents), device="cuda", dtype=torch.int8)
    for _ in range(50):
        env.step(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        env.step(acts)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 2000 * 1e6
    nblk = (batch + env_epw - 1) // env_epw if (env_epw := int(os.environ.get("PGX_EPW", "0"))) else batch
    buf = np.zeros((batch, 4), dtype=np.uint64)
    _lib.check(lib.pgx_debug_timestamps(env._handle, buf.ctypes.data, buf.size))
    buf = buf[buf[:, 0] != 0]
    t = (buf.astype(np.int64) - int(buf[:, 0].min())) / 100.0
    print(f"{name} flags={flags}: {len(buf)} workgroups, {per:.2f} us launch to launch (stamps on)")
    for lab, col in zip(labels, range(4)):
        q = np.percentile(t[:, col], [0, 10, 50, 90, 100])
        print(f"  {lab:32s} min/p10/p50/p90/max = " + " / ".join(f"{v:7.2f}" for v in q))
    d = t[:, 3] - t[:, 0]
    print(f"  per-workgroup lifetime (col3-col0) p50/max = {np.percentile(d,50):.2f} / {d.max():.2f}")
    if flags == 4:  # how many workgroups are alive (started, stores not yet acknowledged) over the launch
        end = t[:, 3].max()
        grid = np.linspace(0, end, 25)
        alive = [(int(((t[:, 0] <= g) & (t[:, 3] > g)).sum())) for g in grid]
        print("  alive workgroups at " + " ".join(f"{g:.0f}us:{a}" for g, a in zip(grid, alive)))
        shares = (C.c_int32 * 8)()
        _lib.check(lib.pgx_xcd_shares(env._handle, shares))
        base = np.concatenate([[0], np.cumsum(list(shares))])
        rot = (base_flags >> 10) & 7
        sl = [((x + rot) & 7) for x in range(8)]  # share written by XCD x
        print(f"  per XCD (rot {rot}; shares {list(shares)}): last stores acknowledged at " +
              " ".join(f"{t[base[s]:base[s + 1], 3].max():.0f}" for s in sl) + " | median " +
              " ".join(f"{np.median(t[base[s]:base[s + 1], 3]):.0f}" for s in sl))
    env.close()
