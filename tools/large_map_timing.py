"""Diagnostic: step time of large maps (the layout that keeps only the occupancy bitmap in LDS, pgx_geometry.multi_wave = 2)
against the same launch on maps that still stage both bitmaps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
for size, agents, r, batch in ((768, 256, 5, 1024), (800, 256, 5, 1024), (1024, 256, 5, 1024), (1024, 256, 7, 1024), (1024, 64, 5, 2048), (1024, 1024, 5, 256)):
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"), batch=batch,
                    auto_reset=True)
    t0 = time.perf_counter()
    env.reset(seed=0)
    torch.cuda.synchronize()
    reset_s = time.perf_counter() - t0
    acts = [torch.randint(0, 5, (batch, agents), device="cuda", dtype=torch.int8) for _ in range(8)]
    for k in range(10):
        env.step(acts[k % 8])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(50):
        env.step(acts[k % 8])
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    W = 2 * r + 1
    obs_bytes = batch * agents * 3 * W * W * 4
    g = env.geometry()
    print(f"{size}x{size} A={agents} r={r} batch={batch}: layout multi_wave={g['multi_wave']} waves={g['waves']} lds={g['lds_bytes']}  "
          f"step {us:8.1f} us = {obs_bytes / us / 1e6:6.2f} TB/s of observations ({obs_bytes / 1e6:.0f} MB), device reset {reset_s:.2f} s", flush=True)
    env.close(release=True)
