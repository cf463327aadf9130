"""Diagnostic: from which map size on is the large-map layout (occupancy bitmap only in LDS, obstacles through the L2; PGX_BIG=1
forces it) faster than staging both bitmaps?  Same instances, same actions, per size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
for size, agents, r, batch in ((256, 256, 7, 4096), (384, 256, 5, 2048), (512, 256, 5, 2048), (640, 256, 5, 1024), (768, 256, 5, 1024), (512, 64, 5, 4096), (768, 64, 5, 2048)):
    line = f"{size}x{size} A={agents} r={r} batch={batch}:"
    for big in ("0", "1"):
        os.environ["PGX_BIG"] = big
        env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"), batch=batch,
                        auto_reset=True, reuse_buffers=True, placement_probe=False)
        env.reset(seed=0)
        acts = [torch.randint(0, 5, (batch, agents), device="cuda", dtype=torch.int8) for _ in range(8)]
        for k in range(10):
            env.step(acts[k % 8])
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(40):
                env.step(acts[k % 8])
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 40 * 1e3)
        g = env.geometry()
        line += f"  {'large-map layout' if big == '1' else 'staged'} (waves {g['waves']}, lds {g['lds_bytes']}): {best:8.1f} us"
        env.close(release=True)
    print(line, flush=True)
