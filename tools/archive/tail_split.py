"""Diagnostic: does a small, low-priority second engine (the last envs, more waves per env) shorten the tail of a step?
One step = engine A (most envs, main stream) + engine B (the rest, low-priority stream), joined on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
K = 300
gc = GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, seed=0, collision_system="soft", max_episode_steps=64)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
def run(nb, waves_b, join=True, prio=True):
    sa = torch.cuda.Stream(priority=-1)
    sb = torch.cuda.Stream(priority=0 if prio else -1)
    envs = []
    for s, n, base, w in ((sa, 8192 - nb, 0, None), (sb, nb, 8192 - nb, waves_b)):
        if n == 0:
            envs.append(None); continue
        if w: os.environ["PGX_WAVES"] = str(w)
        else: os.environ.pop("PGX_WAVES", None)
        with torch.cuda.stream(s):
            e = VecPogema(gc, batch=n, auto_reset=True, reuse_buffers=True, env_index_base=base)
            e.reset(seed=0)
            envs.append((e, torch.randint(0, 5, (n, 64), device="cuda", dtype=torch.int8)))
    os.environ.pop("PGX_WAVES", None)
    torch.cuda.synchronize()
    def step():
        for s, ea in ((sa, envs[0]), (sb, envs[1])):
            if ea is None: continue
            with torch.cuda.stream(s):
                ea[0].step(ea[1])
        if join and envs[1] is not None:
            ea_, eb_ = sa.record_event(), sb.record_event()
            sa.wait_event(eb_); sb.wait_event(ea_)
    for _ in range(30): step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    for ea in envs:
        if ea is not None: ea[0].close()
    return best
print("one engine, 8192 envs            : %.1f us" % run(0, None))
for nb, w in ((1024, 6), (1024, 3), (512, 6), (2048, 3)):
    print(f"A {8192 - nb} + B {nb} ({w} waves/env), low-prio B, joined: %.1f us" % run(nb, w))
print("A 7168 + B 1024 (6 waves), same prio, joined       : %.1f us" % run(1024, 6, prio=False))
