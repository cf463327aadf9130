"""Diagnostic: in-process, interleaved A/B of engine tuning knobs (PGX_FLAGS / PGX_EPW are read at pgx_create),
so that box-to-box and warm-up drift cancel.  usage: python tools/ab_inproc.py cfg2 "PGX_FLAGS=0" "PGX_FLAGS=8"        (a variant may also name another build of
the library: "PGX_LIB=pogema_amd/libpogema_amd_plain.so") """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pogema_amd import GridConfig, VecPogema  # noqa: E402

WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7),
      "a8big": (65536, 16, 8, 5), "a32": (8192, 32, 32, 5), "a32big": (32768, 32, 32, 5), "a24small": (512, 32, 24, 5), "half": (4096, 64, 64, 5), "three_q": (6144, 64, 64, 5), "quarter": (2048, 64, 64, 5), "eighth": (1024, 64, 64, 5)}
wl = sys.argv[1]
fmt = wl.split(":")[1] if ":" in wl else "float32"   # cfg2:u8, cfg2:bfloat16, cfg2:float16
odt = {"float32": torch.float32, "u8": torch.uint8, "uint8": torch.uint8, "bfloat16": torch.bfloat16, "float16": torch.float16}[fmt]
wl = wl.split(":")[0]
variants = sys.argv[2:]
batch, size, agents, r = WL[wl]
envs = []
from pogema_amd import _lib as _L0  # noqa: E402
DEFAULT_LIB = _L0.LIB_PATH
for v in variants:
    for k in ("PGX_FLAGS", "PGX_EPW", "PGX_STAGGER", "PGX_LDS_MIN", "PGX_WAVES", "PGX_STORE", "PGX_TEAM", "PGX_STATE_STORES", "PGX_GATE_NS", "PGX_XCD_SKEW"):
        os.environ.pop(k, None)
    lib_path = None
    for kv in v.split(","):
        if kv:
            k, val = kv.split("=")
            if k == "PGX_LIB":  # another BUILD of the engine in the same process (e.g. make OVL=0 OUT=../libpogema_amd_plain.so)
                lib_path = os.path.abspath(val)
            else:
                os.environ[k] = val.replace(":", ",")  # (list values are written with ':' here: ',' separates the settings)
    from pogema_amd import _lib as _L
    want = lib_path or DEFAULT_LIB
    if _L.LIB_PATH != want or _L._lib is None:
        _L._lib, _L.LIB_PATH = None, want  # VecPogema keeps the library it was created with (self._lib)
    env = VecPogema(GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft"),
                    batch=batch, auto_reset=True, reuse_buffers=True, obs_dtype=odt,
                    placement_probe=False if os.environ.get("AB_PLAIN_BUFFERS") == "1" else None)  # AB_PLAIN_BUFFERS=1: torch-placed buffers
    env.reset(seed=0)
    # every variant writes into the SAME pair of observation buffers: buffer placement alone moves the kernel by up to
    # 10 % (profiles/r1/placement_tiers.txt), which would otherwise drown the effect under test
    if envs:
        env._bufs = [(envs[0]._bufs[k][0],) + env._alloc_outputs(False)[1:] for k in range(2)]
        if batch >= 2048 and os.environ.get("AB_PLAIN_BUFFERS") != "1":  # like the first variant: its own XCD shares, tuned on the shared buffers
            env.tune_xcd_shares(env._bufs[0][0], env._bufs[1][0])
    else:
        env._outputs()
        print("buffers:", getattr(env, "placement", None))
    envs.append(env)
acts = [torch.randint(0, 5, (batch, agents), device="cuda") for _ in range(8)]
rounds, steps = 12, 60
times = np.zeros((len(envs), rounds))
for rd in range(rounds):
    for i, env in enumerate(envs):
        for k in range(5):
            env.step(acts[k % 8])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(steps):
            env.step(acts[k % 8])
        e1.record()
        torch.cuda.synchronize()
        times[i, rd] = e0.elapsed_time(e1) / steps * 1e3
# floor: observation-only passes (no actions, no collision resolve, no state stores) into the same two buffers
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bufs = [envs[0]._bufs[0][0], envs[0]._bufs[1][0]]
for k in range(5):
    envs[0].observe(out=bufs[k & 1])
e0.record()
for k in range(60):
    envs[0].observe(out=bufs[k & 1])
e1.record()
torch.cuda.synchronize()
print(f"{wl} observe-only into the same buffers: {e0.elapsed_time(e1) / 60 * 1e3:8.2f} us")
for v, t in zip(variants, times):
    print(f"{wl} {v:24s} median {np.median(t):8.2f} us  min {t.min():8.2f}  max {t.max():8.2f}")
