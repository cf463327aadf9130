"""Diagnostic: per-workgroup clock stamps of one pgx::step_kernel launch (PGX_FLAGS=4).
Prints when waves start / finish resolving / start storing / end, relative to the first wave, in us
(wall_clock64 ticks at 100 MHz on gfx950), and the number of waves in their store phase over time."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["PGX_FLAGS"] = str(int(os.environ.get("PGX_FLAGS", "0")) | 4)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pogema_amd import GridConfig, VecPogema, _lib  # noqa: E402

WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
batch, size, agents, r = WL[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                batch=batch, auto_reset=True, reuse_buffers=True)
env.reset(seed=0)
acts = torch.randint(0, 5, (batch, agents), device="cuda")
for _ in range(5):
    env.step(acts)
torch.cuda.synchronize()
lib = _lib.load()
lib.pgx_debug_timestamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
buf = np.zeros((batch, 4), dtype=np.uint64)  # one record per workgroup (<= batch)
_lib.check(lib.pgx_debug_timestamps(env._handle, buf.ctypes.data, buf.size))
buf = buf[buf[:, 0] != 0]
print(f"{len(buf)} workgroups")
t = (buf.astype(np.int64) - int(buf[:, 0].min())) / 100.0  # us
print("kernel span (first start -> last end): %.1f us" % t[:, 3].max())
for name, col in (("start", 0), ("resolve done", 1), ("first store", 2), ("end", 3)):
    q = np.percentile(t[:, col], [0, 10, 50, 90, 100])
    print(f"{name:13s} min/p10/p50/p90/max = " + " / ".join(f"{v:7.1f}" for v in q))
print("per-wave prologue (start->first store) p50 %.1f us, p90 %.1f us; store phase p50 %.1f us, p90 %.1f us" % (
    np.percentile(t[:, 2] - t[:, 0], 50), np.percentile(t[:, 2] - t[:, 0], 90),
    np.percentile(t[:, 3] - t[:, 2], 50), np.percentile(t[:, 3] - t[:, 2], 90)))
# estimated write rate over time: every workgroup writes its slice at a uniform rate between its first store and the
# acknowledgement of its last one
wg_bytes = batch * agents * 3 * (2 * r + 1) ** 2 * 4 / len(buf)
dur = np.maximum(t[:, 3] - t[:, 2], 0.01)
step_us = float(os.environ.get("TL_STEP", "4"))
edges = np.arange(0, t[:, 3].max() + step_us, step_us)
print("time(us)  running  storing  est. TB/s over the next %.0f us" % step_us)
for a in edges:
    running = int(((t[:, 0] <= a) & (t[:, 3] > a)).sum())
    storing = int(((t[:, 2] <= a) & (t[:, 3] > a)).sum())
    ov = np.clip(np.minimum(t[:, 3], a + step_us) - np.maximum(t[:, 2], a), 0, None)
    rate = float((ov / dur * wg_bytes).sum()) / (step_us * 1e-6) / 1e12
    print(f"{a:7.0f}  {running:7d}  {storing:7d}  {rate:6.2f}")
