"""Long soak of the on-device loop (round 6): engine A runs `pgx_rollout` launches of K steps (K = 64, 37, 200, 8, 1 ... --
action blocks of eight end inside, at and across launch boundaries), engine B the same actions through step(); after
every launch every per-step output of all environments (rewards, flags, is_active, episode_done, metrics, the last
observations) and the complete state must be identical.  BASELINE shapes at full size, all collision systems / modes.
usage: python tools/soak_rollout.py [total steps per case = 3000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pogema_amd import GridConfig, VecPogema
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
CASES = [("cfg2", 8192, 64, 64, 5, "soft", "finish"), ("cfg3", 8192, 32, 16, 5, "priority", "restart"), ("cfg1", 1024, 16, 8, 5, "block_both", "nothing"),
         ("cfg4", 1024, 256, 256, 7, "soft", "restart"), ("cfg3b", 8192, 32, 16, 5, "soft", "finish"), ("big", 64, 1024, 100, 5, "soft", "finish")]
KS = [64, 37, 200, 8, 1, 9, 129]
for name, B, size, A, r, coll, ont in CASES:
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, density=0.3, seed=3, collision_system=coll, on_target=ont, max_episode_steps=48)
    a = VecPogema(gc, batch=B, auto_reset=True)
    b = VecPogema(gc, batch=B, auto_reset=True)
    a.reset(seed=3)
    b.reset(seed=3)
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    done, launches, t0, ok = 0, 0, time.time(), True
    steps_case = T if size <= 256 else min(T, 400)
    while done < steps_case and ok:
        K = KS[launches % len(KS)]
        acts = torch.randint(0, 5, (K, B, A), generator=gen, device="cuda", dtype=torch.int8)
        out = a.rollout(acts, obs_slots=2)
        ref = {k: [] for k in ("rewards", "terminated", "truncated", "is_active", "episode_done", "metrics")}
        for t in range(K):
            obs, rew, term, trunc, infos = b.step(acts[t])
            ref["rewards"].append(rew.clone()); ref["terminated"].append(term.clone()); ref["truncated"].append(trunc.clone())
            ref["is_active"].append(infos["is_active"].clone()); ref["episode_done"].append(infos["episode_done"].clone())
            ref["metrics"].append(torch.where(infos["episode_done"][:, None], infos["metrics"], torch.zeros_like(infos["metrics"])))
        for k in ref:
            got = out[k] if k != "metrics" else torch.where(out["episode_done"][..., None], out["metrics"], torch.zeros_like(out["metrics"]))
            if not torch.equal(torch.stack(ref[k]), got):
                print(f"MISMATCH {name}: {k} in launch {launches} (K={K}, steps {done}..{done + K})"); ok = False
        if not torch.equal(out["obs"][(K - 1) % 2], obs):
            print(f"MISMATCH {name}: last observation of launch {launches}"); ok = False
        sa, sb = a.get_state(), b.get_state()
        for k in sa:
            if not torch.equal(sa[k], sb[k]):
                print(f"MISMATCH {name}: state {k} after launch {launches}"); ok = False
        del out
        done += K
        launches += 1
    g = a.geometry(for_rollout=True)
    print(f"{name} {B} envs {size}x{size} A={A} {coll}/{ont}: {done} steps in {launches} rollout launches (K in {KS}) == step() loop, outputs and state: "
          f"{'OK' if ok else 'FAILED'} ({time.time() - t0:.0f} s; rollout geometry lanes {g['lanes_per_env']} envs/wave {g['envs_per_wave']} waves {g['waves']} layout {g['multi_wave']})", flush=True)
    a.close(); b.close()
    if not ok:
        sys.exit(1)
