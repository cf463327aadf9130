"""Diagnostic: the placement probe's view -- observation-pass time of every candidate buffer of a configs[2] env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
env = VecPogema(GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, seed=0, collision_system="soft"),
                batch=8192, auto_reset=True, reuse_buffers=True)
env.reset(seed=0)
env.step(torch.zeros((8192, 64), dtype=torch.int64, device="cuda"))
print("placement candidates (us):", env.placement_us)
