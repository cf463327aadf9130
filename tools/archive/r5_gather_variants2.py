"""One-off (round 5): isolate the host-gather cost.  A: free-running side-stream copies; B: + event dependency; C: + host
sync at depth 2; D: zero-copy -- the step kernel writes its small outputs straight into pinned host memory."""
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
    print(f"{label:70s} {statistics.median(out):8.1f} us/step", flush=True)

def plain(k):
    for i in range(k):
        env.step(pool[i % 16])
timed(plain, "plain loop")

def block_of(out):
    rew = out[1]
    st = rew.untyped_storage()
    return torch.empty(0, dtype=torch.uint8, device=rew.device).set_(st, rew.data_ptr() - st.data_ptr(), (7 * n,))

def A_free(k):
    for i in range(k):
        out = env.step(pool[i % 16])
        src = block_of(out)
        with torch.cuda.stream(side):
            host[i % 4].copy_(src, non_blocking=True)
timed(A_free, "A: copy on side stream, NO dependency, no host sync")

def B_dep(k):
    for i in range(k):
        out = env.step(pool[i % 16])
        src = block_of(out)
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            host[i % 4].copy_(src, non_blocking=True)
timed(B_dep, "B: + wait_event on the step")

def C_sync(k, depth=2):
    evs = []
    for i in range(k):
        out = env.step(pool[i % 16])
        src = block_of(out)
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            host[i % 4].copy_(src, non_blocking=True)
            d = torch.cuda.Event(); d.record(side)
        evs.append(d)
        if len(evs) > depth:
            evs.pop(0).synchronize()
timed(C_sync, "C: + host waits for the copy of step t-2")
timed(lambda k: C_sync(k, 1), "C1: + host waits for the copy of step t-1")

def C_same_stream(k):
    for i in range(k):
        out = env.step(pool[i % 16])
        src = block_of(out)
        host[i % 4].copy_(src, non_blocking=True)
timed(C_same_stream, "E: copy on the SAME stream as the steps (serial by construction)")

# D: zero copy
obs_bufs = [torch.empty(env.obs_shape, dtype=torch.float32, device="cuda") for _ in range(2)]
hsets = []
for s in range(4):
    blk = host[s]
    hsets.append((blk[:4 * n].view(torch.float32).view(B, A), blk[4 * n:5 * n].view(torch.bool).view(B, A),
                  blk[5 * n:6 * n].view(torch.bool).view(B, A), blk[6 * n:7 * n].view(torch.bool).view(B, A)))
env._check_out = lambda out: out  # (experiment only)
def D_zero(k):
    for i in range(k):
        h = hsets[i % 4]
        env.step(pool[i % 16], out=(obs_bufs[i % 2], h[0], h[1], h[2], h[3]))
try:
    timed(D_zero, "D: zero copy -- kernel writes rewards/flags into pinned host memory")
    # correctness: same step on device outputs
    env2 = VecPogema(env.grid_config, batch=B, auto_reset=True, reuse_buffers=False, placement_budget_gib=0)
    env2.reset(seed=0)
    env3 = VecPogema(env.grid_config, batch=B, auto_reset=True, reuse_buffers=False, placement_budget_gib=0)
    env3.reset(seed=0)
    env3._check_out = lambda out: out
    ok = True
    for i in range(20):
        a = env2.step(pool[i % 16])
        h = hsets[i % 4]
        env3.step(pool[i % 16], out=(obs_bufs[0], h[0], h[1], h[2], h[3]))
        torch.cuda.synchronize()
        ok &= torch.equal(a[1].cpu(), h[0]) and torch.equal(a[2].cpu(), h[1]) and torch.equal(a[3].cpu(), h[2]) and torch.equal(a[4]["is_active"].cpu(), h[3])
    print("D correctness over 20 steps:", ok, "rewards sum", float(h[0].sum()))
except Exception as exc:
    print("D failed:", repr(exc))
