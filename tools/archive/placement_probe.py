"""Diagnostic: does the speed of the observation stream depend on WHERE the output buffer lives?
One env, several candidate obs tensors, the same kernel (pgx_observe / pgx_step) into each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pogema_amd import GridConfig, VecPogema, _lib

batch, size, agents, r = 8192, 64, 64, 5
env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                batch=batch, auto_reset=True)
env.reset(seed=0)
acts = torch.randint(0, 5, (batch, agents), device="cuda")
rew = torch.empty((batch, agents), device="cuda"); te = torch.empty((batch, agents), dtype=torch.bool, device="cuda")
tr = torch.empty_like(te); ac = torch.empty_like(te)
bufs = [torch.empty(env.obs_shape, dtype=torch.float32, device="cuda") for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
def time_into(buf, n=40):
    for _ in range(5):
        _lib.check(env._lib.pgx_step(env._handle, acts.data_ptr(), 2, buf.data_ptr(), rew.data_ptr(), te.data_ptr(), tr.data_ptr(), ac.data_ptr(), env._stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.check(env._lib.pgx_step(env._handle, acts.data_ptr(), 2, buf.data_ptr(), rew.data_ptr(), te.data_ptr(), tr.data_ptr(), ac.data_ptr(), env._stream()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(2):
    for i, b in enumerate(bufs):
        p = b.data_ptr()
        print(f"round {rnd} buf {i} ptr {p:#x} mod2M {p % (2<<20):#x} mod1G {p % (1<<30):#x}: {time_into(b):7.2f} us")

if len(sys.argv) > 2: sys.exit(0)
# ---- part 2: the same physical allocation, different offsets ----
n_obs = int(np.prod(env.obs_shape))
big = torch.empty(n_obs + (600 << 20) // 4, dtype=torch.float32, device="cuda")
print(f"big ptr {big.data_ptr():#x}")
for off_bytes in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 64 << 20, 256 << 20, 512 << 20):
    view = big[off_bytes // 4: off_bytes // 4 + n_obs].view(env.obs_shape)
    print(f"offset {off_bytes:>10d}: {time_into(view):7.2f} us")
