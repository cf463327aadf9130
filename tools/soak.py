"""Long soak at the headline size: the engine steps ALL 8192 configs[2] environments (three waves per environment, tuned
XCD shares, reused buffers, auto-reset) for T steps; a sample of them is stepped by the plain-C oracle on the host with
the same actions, and every observation / reward / flag of the sample is compared at every step.  Also one lifelong run.
usage: python tools/soak.py [T=1500] [sample=48]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle.c_oracle import COracle
from pogema_amd import GridConfig, VecPogema

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
S = int(sys.argv[2]) if len(sys.argv) > 2 else 48
B, A, size, r = 8192, 64, 64, 5
for on_target, collision in (("finish", "soft"), ("restart", "priority"), ("nothing", "block_both")):
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, density=0.3, seed=0, collision_system=collision, on_target=on_target,
                    max_episode_steps=48)
    env = VecPogema(gc, batch=B, auto_reset=True)  # product default: recycled output sets
    obs, _ = env.reset(seed=0)
    obstacles, agents, targets = (v.cpu().numpy() for v in env._initial)
    st = env.get_state()
    rng = np.random.default_rng(1)
    idx = np.sort(rng.choice(B, S, replace=False))
    refs = []
    for b in idx:  # one single-env oracle per sampled env, keyed by its global index (lifelong stream)
        o = COracle(1, size, size, A, r, collision, on_target, 48, True, 0, int(b))
        o.reset(obstacles[b:b + 1], st["agents_xy"][b:b + 1].cpu().numpy(), st["targets_xy"][b:b + 1].cpu().numpy())
        refs.append(o)
    didx = torch.as_tensor(idx, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    t0 = time.time(); bad = 0
    for t in range(T):
        act = torch.randint(0, 5, (B, A), generator=gen, device="cuda", dtype=torch.int8)
        obs, rew, term, trunc, infos = env.step(act)
        a_h = act[didx].cpu().numpy().astype(np.int64)
        o_h, r_h, te_h, tr_h = obs[didx].cpu().numpy(), rew[didx].cpu().numpy(), term[didx].cpu().numpy(), trunc[didx].cpu().numpy()
        for k, o in enumerate(refs):
            ro, rr, rte, rtr, _ = o.step(a_h[k:k + 1])
            if not (np.array_equal(ro[0], o_h[k]) and np.allclose(rr[0], r_h[k], atol=1e-6) and np.array_equal(rte[0].astype(bool), te_h[k].astype(bool))
                    and np.array_equal(rtr[0].astype(bool), tr_h[k].astype(bool))):
                bad += 1
                print(f"MISMATCH {on_target}/{collision} step {t} env {idx[k]}"); break
        if not bad and t % 97 == 96:  # the occupancy array is STATE: observe() shows what the step showed (docs/SPEC.md Q2)
            if not torch.equal(env.observe()[didx], obs[didx]):
                bad += 1
                print(f"MISMATCH {on_target}/{collision} step {t}: observe() after the step differs from the step's observation")
        if bad: break
    for o in refs: o.close()
    print(f"{on_target}/{collision}: {T} steps x {B} envs on the engine, {S} sampled envs checked at every step against the C oracle: "
          f"{'OK' if not bad else 'FAILED'} ({time.time() - t0:.0f} s); placement {env.placement.get('xcd_shares')}", flush=True)
    env.close()
    if bad: sys.exit(1)
