"""Diagnostic: one million environments in one engine (BASELINE configs[0] geometry: 8x8, 2 agents, r = 3) against the C
oracle, every output of every environment for T steps -- index arithmetic at batch sizes far beyond the benchmark's.
usage: python tools/huge_batch_check.py [log2_batch=20] [T=12]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from util import assert_rollouts_equal, c_oracle_rollout, engine_rollout, generate_instances, random_actions

B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for (H, A, r, collision, on_target) in ((8, 2, 3, "priority", "finish"), (8, 2, 3, "soft", "restart"), (12, 5, 2, "block_both", "nothing")):
    t0 = time.time()
    obstacles, agents, targets = generate_instances(B, H, H, A, 0.3, seed=1)
    actions = random_actions(T, B, A, seed=2)
    kw = dict(obs_radius=r, collision_system=collision, on_target=on_target, max_episode_steps=8, auto_reset=True, seed=5,
              env_index_base=(1 << 33) + 7)  # global indices beyond 32 bits as well
    ref = c_oracle_rollout(obstacles, agents, targets, actions, nthreads=32, **kw)
    got = engine_rollout(obstacles, agents, targets, actions, action_dtype="int8", **kw)
    assert_rollouts_equal(ref, got, f"huge batch {B} x {A}")
    print(f"{B} envs x {A} agents ({H}x{H}, r={r}, {collision}/{on_target}): {T} steps identical with the C oracle "
          f"({time.time() - t0:.0f} s)", flush=True)
