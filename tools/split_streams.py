"""Diagnostic: one batch as S independent engines on S streams (double-buffered sampling: the halves' launch boundaries
overlap with each other's observation streams) against one engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
WL = {"cfg1": (1024, 16, 8, 5), "cfg2": (8192, 64, 64, 5), "cfg3": (8192, 32, 16, 5), "cfg4": (4096, 256, 256, 7)}
K = int(os.environ.get("K", "300"))
for name in sys.argv[1:] or ["cfg2"]:
    batch, size, agents, r = WL[name]
    gc = GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.3, seed=0, collision_system="soft", max_episode_steps=64)
    line = f"{name}:"
    probe = os.environ.get("PROBE", "1") == "1"
    for S in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        envs, acts = [], []
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                e = VecPogema(gc, batch=batch // S, auto_reset=True, reuse_buffers=True, env_index_base=s * (batch // S),
                              placement_probe=probe)
                e.reset(seed=0)
                envs.append(e)
                acts.append(torch.randint(0, 5, (batch // S, agents), device="cuda", dtype=torch.int8))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            join = os.environ.get("JOIN", "0") == "1"
            for t in range(K):
                for s in range(S):
                    with torch.cuda.stream(streams[s]):
                        envs[s].step(acts[s])
                if join and S > 1:  # lockstep: nobody starts step t+1 before everybody finished step t (device-side join)
                    evs = [streams[s].record_event() for s in range(S)]
                    for s in range(S):
                        for o in range(S):
                            if o != s:
                                streams[s].wait_event(evs[o])
            torch.cuda.synchronize()
            if rep:
                best = min(best, (time.perf_counter() - t0) / K * 1e6)
        pl = [(e.placement or {}) for e in envs]
        line += f"  {S} engine(s) {best:8.2f} us per full-batch step [" + ",".join(
            f"{'S' if q.get('spread') else '-'}{q.get('spacer_gib', 0):.0f}" for q in pl) + "]"
        for e in envs:
            e.close()
        del envs, acts
        torch.cuda.empty_cache()
    print(line, flush=True)
