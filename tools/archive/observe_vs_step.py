"""Diagnostic: step() vs observe-only passes (no actions, no collision resolve, no state stores) into the SAME two
alternating observation buffers -- separates what the state phase costs from what the stream costs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg3x8": (65536, 32, 16, 5),
      "cfg1x64": (65536, 16, 8, 5)}
for wl in sys.argv[1:]:
    batch, size, agents, r = WL[wl]
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                    batch=batch, auto_reset=True, reuse_buffers=True, placement_probe=False)
    env.reset(seed=0)
    acts = [torch.randint(0, 5, (batch, agents), device="cuda", dtype=torch.int8) for _ in range(8)]
    env.step(acts[0])
    bufs = [env._bufs[0][0], env._bufs[1][0]]
    def timed(fn, n=200):
        for k in range(10): fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n): fn(k)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res = {}
    for rep in range(3):
        res.setdefault("step", []).append(timed(lambda k: env.step(acts[k % 8])))
        res.setdefault("step_noobs", []).append(timed(lambda k: env.step(acts[k % 8], compute_obs=False)))
        res.setdefault("observe_2buf", []).append(timed(lambda k: env.observe(out=bufs[k & 1])))
        res.setdefault("observe_1buf", []).append(timed(lambda k: env.observe(out=bufs[0])))
    mb = np.prod(env.obs_shape) * 4 / 1e6
    print(f"{wl}: obs {mb:.0f} MB  " + "  ".join(f"{k} {np.median(v):.2f} us ({mb / np.median(v) / 1e3 * 1e3:.0f} GB/s)" if "noobs" not in k else f"{k} {np.median(v):.2f} us" for k, v in res.items()), flush=True)
    env.close(); del env, bufs
    torch.cuda.empty_cache()
