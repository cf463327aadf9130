"""Diagnostic: K steps as one pgx_rollout launch against K pgx_step launches (same buffers' sizes, two observation slots)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
K = int(os.environ.get("K", "200"))
for name in sys.argv[1:] or ["cfg1", "cfg3", "cfg2"]:
    batch, size, agents, r = WL[name]
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft",
                               max_episode_steps=64), batch=batch, auto_reset=True, reuse_buffers=True)
    env.reset(seed=0)
    acts = torch.randint(0, 5, (K, batch, agents), device="cuda", dtype=torch.int8)
    for t in range(20):
        env.step(acts[t])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(0 if os.environ.get("NO_STEP_LOOP") else 3):
        t0 = time.perf_counter()
        for t in range(K):
            env.step(acts[t])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    line = f"{name}: step loop {best:8.2f} us/step"
    for slots in [int(v) for v in os.environ.get("SLOTS", "2,1,0").split(",")]:
        env.rollout(acts[:4], obs_slots=slots)
        torch.cuda.synchronize()
        rb = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = env.rollout(acts, obs_slots=slots)
            b.record()
            torch.cuda.synchronize()
            rb = min(rb, a.elapsed_time(b) * 1e3 / K)
            del out
        q = env.placement or {}
        line += f" | rollout({slots} slots) {rb:8.2f} [{'S' if q.get('spread') else '-'}{q.get('spacer_gib', 0):.0f}]"
    obs_bytes = batch * agents * 3 * (2 * r + 1) ** 2 * 4
    print(line + f" | obs {obs_bytes / 1e6:.0f} MB -> 8 TB/s = {obs_bytes / 8e12 * 1e6:.1f} us", flush=True)
    env.close()
