import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogema_amd import GridConfig, pogema_v0
for A in (2, 8, 64):
    env = pogema_v0(GridConfig(size=16 if A < 64 else 64, num_agents=A, obs_radius=5, density=0.3, seed=0, max_episode_steps=10**6))
    env.reset(seed=0)
    acts = [env.sample_actions() for _ in range(64)]
    for i in range(50): env.step(acts[i % 64])
    t0 = time.perf_counter(); n = 1000
    for i in range(n): env.step(acts[i % 64])
    dt = time.perf_counter() - t0
    print(f"list API, {A} agents: {dt/n*1e6:.1f} us/step = {n/dt:.0f} steps/s")
    env.close()
