"""Diagnostic (round 3): where the host time of one VecPogema.step() goes (per-component microbenchmarks + cProfile), on a
small environment whose kernel is shorter than the Python call.  usage: python tools/host_profile.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
gc = GridConfig(size=16, num_agents=8, obs_radius=5, density=0.3, seed=0, collision_system="soft")
env = VecPogema(gc, batch=1024, auto_reset=True)
env.reset(seed=0)
acts = torch.randint(0, 5, (1024, 8), device="cuda", dtype=torch.int8)
for _ in range(100): o = env.step(acts)
torch.cuda.synchronize()
def t(f, n=20000):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("stream()", t(env._stream))
print("raw stream", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("capturing?", t(torch.cuda.is_current_stream_capturing))
print("prepare_actions", t(lambda: env._prepare_actions(acts)))
print("recycled", t(lambda: env._recycled(True)))
print("data_ptr", t(acts.data_ptr))
print("bad_action getattr", t(lambda: env.semantics.bad_action == "flag"))
pr = cProfile.Profile()
pr.enable()
for _ in range(20000): o = env.step(acts)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
