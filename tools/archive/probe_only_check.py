"""What a default-constructed VecPogema does on a device that is NOT its own (here: 40 GiB of it already taken by this very
process): the probe-only placement of round 4 -- nothing held, one probe pair -- and what the configs[2] step then costs,
next to fresh torch tensors per step and an explicit walk.   python tools/archive/probe_only_check.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pogema_amd import GridConfig, VecPogema  # noqa: E402

ballast = torch.empty(40 << 30, dtype=torch.uint8, device="cuda")  # somebody else's memory: < 90 % of the device is free
gc = GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, seed=0, collision_system="soft")
acts = [torch.randint(0, 5, (8192, 64), device="cuda", dtype=torch.int8) for _ in range(8)]
for name, kw in (("default (shared device -> probe only)", {}), ("fresh torch tensors per step", {"reuse_buffers": False}),
                 ("explicit walk, half of the free memory", {"placement_budget_gib": "half"})):
    env = VecPogema(gc, batch=8192, auto_reset=True, **kw)
    t0 = time.perf_counter()
    env.reset(seed=0)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    for k in range(50):
        env.step(acts[k % 8])
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(300):
            env.step(acts[k % 8])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
    pl = env.placement or {}
    print(f"{name:42s} {best:6.1f} us per step; reset() incl. buffer set-up {setup * 1e3:7.1f} ms; spread={pl.get('spread')} "
          f"candidates={pl.get('candidates')} held={pl.get('spacer_gib')} policy: {str(pl.get('policy'))[:70]}")
    env.close(release=True)
