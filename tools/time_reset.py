"""Diagnostic: wall time of the on-device reset vs the host generator + upload, BASELINE geometries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
for name, B, S, A, r, ot in [("cfg2", 8192, 64, 64, 5, "finish"), ("cfg2 lifelong", 8192, 64, 64, 5, "restart"),
                             ("cfg3", 8192, 32, 16, 5, "finish"), ("cfg4", 4096, 256, 256, 7, "finish"),
                             ("cfg4 lifelong", 4096, 256, 256, 7, "restart"), ("cfg1", 1024, 16, 8, 5, "finish")]:
    env = VecPogema(GridConfig(size=S, num_agents=A, obs_radius=r, density=0.3, seed=0, on_target=ot), batch=B)
    env.reset(seed=0); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.reset(seed=1); torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); o, a, t = env.generate(2); t_gen = time.perf_counter() - t0
    t0 = time.perf_counter(); env.reset_from_state(o, a, t, validate=False); torch.cuda.synchronize(); t_up = time.perf_counter() - t0
    mask = torch.zeros(B, dtype=torch.bool, device="cuda"); mask[::10] = True
    t0 = time.perf_counter(); env.reset_where(mask); torch.cuda.synchronize(); t_mask = time.perf_counter() - t0
    print(f"{name:14s} device reset {t_dev*1e3:8.2f} ms | host generate {t_gen*1e3:8.2f} ms + install {t_up*1e3:8.2f} ms | "
          f"regenerate 10% {t_mask*1e3:8.2f} ms", flush=True)
    env.close()
