"""One-off (round 5): where does the host-gather cost per step come from?  Variants of the pipelined loop on configs[2]."""
import sys, time, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pogema_amd import GridConfig, VecPogema
from pogema_amd.sharding import HostGather, start_step_gather, step_output_fields

B, A = 8192, 64
env = VecPogema(GridConfig(size=64, density=0.3, num_agents=A, obs_radius=5, seed=0, collision_system="soft", max_episode_steps=64),
                batch=B, auto_reset=True)
env.reset(seed=0)
env.warm_buffers()
pool = [torch.randint(0, 5, (B, A), device="cuda").to(torch.int8) for _ in range(16)]
N = 400

def timed(fn, label):
    fn(40)
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(N); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / N * 1e6)
    print(f"{label:60s} {statistics.median(out):8.1f} us/step", flush=True)

def plain(n):
    for i in range(n):
        env.step(pool[i % 16])
timed(plain, "plain loop")

def make(depth, slots, with_metrics=True, clone=True):
    g = HostGather(step_output_fields(env, with_metrics=with_metrics), B, slots=slots)
    def loop(n):
        pend = []
        for i in range(n):
            out = env.step(pool[i % 16])
            if clone or not with_metrics:
                tk = start_step_gather(g, out)
            else:
                obs, rew, term, trunc, infos = out
                tk = g.start(rewards=rew, terminated=term, truncated=trunc, is_active=infos["is_active"],
                             episode_done=infos["episode_done"], metrics=infos["metrics"])
            del out
            pend.append(tk)
            if len(pend) > depth:
                g.finish(pend.pop(0))
        while pend:
            g.finish(pend.pop(0))
    return loop

timed(make(1, 3), "depth 1, 3 slots, metrics+clone (bench form)")
timed(make(2, 4), "depth 2, 4 slots, metrics+clone")
timed(make(3, 5), "depth 3, 5 slots, metrics+clone")
timed(make(1, 3, with_metrics=False), "depth 1, only the 7 B/agent block")
timed(make(2, 4, with_metrics=False), "depth 2, only the 7 B/agent block")
timed(make(2, 4, clone=False), "depth 2, metrics without the snapshot clones (racy)")

# raw: how long does one 3.67 MB D2H take while the step kernel runs / while idle
host = torch.empty(7 * B * A, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(7 * B * A, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
for busy in (False, True):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if busy:
        for i in range(40):
            env.step(pool[i % 16])
    with torch.cuda.stream(side):
        e0.record(side)
        for _ in range(10):
            host.copy_(dev, non_blocking=True)
        e1.record(side)
    torch.cuda.synchronize()
    print(f"3.67 MB D2H x10 {'under the step stream' if busy else 'idle GPU'}: {e0.elapsed_time(e1) * 100:.1f} us each", flush=True)
env.close()
