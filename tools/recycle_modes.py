"""Diagnostic (round 3): step time and host enqueue time per step() of the buffer modes, one workload.
usage: python tools/recycle_modes.py cfg3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
batch, size, agents, r = WL[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
gc = GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft")
acts = [torch.randint(0, 5, (batch, agents), device="cuda", dtype=torch.int8) for _ in range(8)]
for name, kw, nrec in (("reuse_buffers=True", dict(reuse_buffers=True), 0), ("recycle (default)", dict(), 0),
                       ("reuse_buffers=False", dict(reuse_buffers=False), 0)):
    env = VecPogema(gc, batch=batch, auto_reset=True, placement_budget_gib="all", **kw)
    env.reset(seed=0)
    env.warm_buffers()
    for k in range(50):
        obs = env.step(acts[k % 8])[0]
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(1000):
            obs = env.step(acts[k % 8])[0]
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / 1000 * 1e6, (t2 - t0) / 1000 * 1e6))
    res.sort(key=lambda x: x[1])
    print(f"{name:22s} host enqueue {res[1][0]:6.1f} us/step   wall {res[1][1]:6.1f} us/step   placement spread={env.placement.get('spread') if env.placement else None}")
    del obs
    env.close()
    del env
    torch.cuda.empty_cache()
