"""One-off (round 5): host-mediated dependency -- the copy of step t is ISSUED only once the host has seen step t complete,
so no cross-stream wait is ever enqueued on the GPU."""
import sys, time, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pogema_amd import GridConfig, VecPogema

B, A = 8192, 64
env = VecPogema(GridConfig(size=64, density=0.3, num_agents=A, obs_radius=5, seed=0, collision_system="soft", max_episode_steps=64),
                batch=B, auto_reset=True)
env.reset(seed=0)
env.warm_buffers()
pool = [torch.randint(0, 5, (B, A), device="cuda").to(torch.int8) for _ in range(16)]
N = 400
n = B * A
host = [torch.empty(7 * n, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
side = torch.cuda.Stream()

def timed(fn, label):
    fn(40)
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(N); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / N * 1e6)
    print(f"{label:75s} {statistics.median(out):8.1f} us/step", flush=True)

def plain(k):
    for i in range(k):
        env.step(pool[i % 16])
timed(plain, "plain loop")

def plain_ev(k):
    for i in range(k):
        env.step(pool[i % 16])
        ev = torch.cuda.Event(); ev.record()
timed(plain_ev, "plain loop + an event recorded after every step")

def block_of(out):
    rew = out[1]
    st = rew.untyped_storage()
    return torch.empty(0, dtype=torch.uint8, device=rew.device).set_(st, rew.data_ptr() - st.data_ptr(), (7 * n,))

def F(k, lag=1, wait_done=2):
    ready, done, srcs = [], [], []
    for i in range(k):
        out = env.step(pool[i % 16])
        srcs.append(block_of(out)); del out
        ev = torch.cuda.Event(); ev.record(); ready.append(ev)
        if i >= lag:
            j = i - lag
            ready[j].synchronize()              # host: step j is complete (step j+1.. are already enqueued)
            with torch.cuda.stream(side):
                host[j % 4].copy_(srcs[j], non_blocking=True)
                d = torch.cuda.Event(); d.record(side)
            done.append(d); srcs[j] = None
            if len(done) > wait_done:
                done[-1 - wait_done].synchronize()
timed(lambda k: F(k, 1, 2), "F: host-mediated, copy t-1 issued after step t enqueued; wait copy t-3")
timed(lambda k: F(k, 1, 0), "F0: same, host waits for copy t-1 at once")
timed(lambda k: F(k, 2, 1), "F2: lag 2 (steps t, t-1 in the queue), wait copy t-3")

def Fq(k):
    """query-driven pump: never block on `ready`; issue whatever is ready; block only on done of t-3"""
    ready, done, srcs, nxt = [], {}, [], 0
    for i in range(k):
        out = env.step(pool[i % 16])
        srcs.append(block_of(out)); del out
        ev = torch.cuda.Event(); ev.record(); ready.append(ev)
        while nxt <= i and ready[nxt].query():
            with torch.cuda.stream(side):
                host[nxt % 4].copy_(srcs[nxt], non_blocking=True)
                d = torch.cuda.Event(); d.record(side)
            done[nxt] = d; srcs[nxt] = None; nxt += 1
        if i >= 3:
            if (i - 3) not in done:
                ready[i - 3].synchronize()
                while nxt <= i - 3:
                    with torch.cuda.stream(side):
                        host[nxt % 4].copy_(srcs[nxt], non_blocking=True)
                        d = torch.cuda.Event(); d.record(side)
                    done[nxt] = d; srcs[nxt] = None; nxt += 1
            done.pop(i - 3).synchronize()
timed(Fq, "Fq: query-driven pump, host blocks only on copy t-3")
