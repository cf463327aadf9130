"""Diagnostic: candidates carved out of ONE large allocation vs separate allocations (placement probe view)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pogema_amd import GridConfig, VecPogema, _lib
env = VecPogema(GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, seed=0, collision_system="soft"),
                batch=8192, auto_reset=True)
env.reset(seed=0)
n_obs = int(np.prod(env.obs_shape))
us = C.c_float()
def probe(t):
    _lib.check(env._lib.pgx_time_observe(env._handle, t.data_ptr(), 3, C.byref(us), env._stream()))
    return round(float(us.value), 1)
for slab_gb in (24,):
    n = int(slab_gb * (1 << 30) // (n_obs * 4))
    slab = torch.empty(n * n_obs, dtype=torch.float32, device="cuda")
    res = [probe(slab[i * n_obs:(i + 1) * n_obs].view(env.obs_shape)) for i in range(n)]
    print(f"one {slab_gb} GB slab at {slab.data_ptr():#x}, {n} slices in address order:", res)
    del slab
sep = [torch.empty(env.obs_shape, dtype=torch.float32, device="cuda") for _ in range(32)]
print("32 separate allocations in allocation order:", [(hex(t.data_ptr()), probe(t)) for t in sep])
