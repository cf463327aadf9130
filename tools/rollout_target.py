"""Profile target: LAUNCHES pgx_rollout launches of K steps each of one workload (rocprofv3 puts the program directly
behind `--`).  usage: python3 tools/rollout_target.py cfg2 SLOTS [K] [LAUNCHES]; prints one JSON line with the HIP-event time."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
name, slots = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = int(sys.argv[4]) if len(sys.argv) > 4 else 10
batch, size, agents, r = WL[name]
env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft", max_episode_steps=64),
                batch=batch, auto_reset=True)
env.reset(seed=0)
acts = torch.randint(0, 5, (K, batch, agents), device="cuda", dtype=torch.int8)
env.rollout(acts, obs_slots=slots)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(L):
    out = env.rollout(acts, obs_slots=slots)
b.record()
torch.cuda.synchronize()
W = 2 * r + 1
bpas = 12.0 * W * W + 3.0 * (((size + 2 * r) ** 2 + 7) // 8) / agents + 21.0
print(json.dumps({"workload": name, "obs_slots": slots, "steps_per_launch": K, "launches": L,
                  "us_per_step_hip_events": a.elapsed_time(b) * 1e3 / (K * L), "us_per_launch_hip_events": a.elapsed_time(b) * 1e3 / L,
                  "algorithmic_bytes_per_step": bpas * batch * agents, "obs_bytes_per_step": batch * agents * 3 * W * W * 4,
                  "ring_mib": slots * batch * agents * 3 * W * W * 4 / 2 ** 20, "placement_spread": bool((env.placement or {}).get("spread")),
                  "geometry": env.geometry(for_rollout=True)}))
