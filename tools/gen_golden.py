#!/usr/bin/env python3
"""Generate REAL reference fixtures -- run this where `import pogema` works.

In this build container it cannot run: /root/reference holds only README.md and neither `pogema` nor
`gymnasium` is installed.  It is committed so that a maintainer with the real package can convert
"parity unpinned" into a pinned contract:

    pip install pogema            # or: PYTHONPATH=/path/to/Cognitive-AI-Systems/pogema
    python tools/gen_golden.py    # writes tests/golden/reference_*.npz

tests/test_golden_reference.py then checks the oracle (CPU) and the HIP engine (GPU) against every
fixture found.  The reference's Python never travels: only the .npz vectors (inputs + expected
outputs) are committed.

What is recorded per case: the initial state actually used by the reference (obstacles, agents_xy,
targets_xy read back from its Grid, unpadded), the action stream, and per step agents_xy, targets_xy,
is_active, rewards, terminated, truncated and the full float32 observations.
"""
import itertools
import os
import sys

import numpy as np


def main():
    try:
        from pogema import GridConfig, pogema_v0
    except Exception as exc:  # pragma: no cover - depends on the environment
        sys.exit(f"pogema is not importable here ({exc!r}); nothing generated")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    geoms = [dict(size=8, num_agents=2, obs_radius=3, density=0.3),      # BASELINE.json configs[0]
             dict(size=16, num_agents=8, obs_radius=5, density=0.3),
             dict(size=32, num_agents=16, obs_radius=5, density=0.3),
             dict(size=12, num_agents=40, obs_radius=2, density=0.1)]    # collision-dense
    n = 0
    for g, cs, ot, seed in itertools.product(geoms, ("priority", "block_both", "soft"),
                                             ("finish", "restart", "nothing"), (0, 1, 2)):
        gc = GridConfig(seed=seed, collision_system=cs, on_target=ot, max_episode_steps=32, **g)
        env = pogema_v0(gc)
        obs, _ = env.reset(seed=seed)
        grid = env.unwrapped.grid if hasattr(env, "unwrapped") else env.grid
        r = gc.obs_radius
        obstacles = np.asarray(grid.get_obstacles(ignore_borders=True), dtype=np.uint8)
        agents0 = np.asarray(grid.get_agents_xy(ignore_borders=True), dtype=np.int32)
        targets0 = np.asarray(grid.get_targets_xy(ignore_borders=True), dtype=np.int32)
        rng = np.random.default_rng(1000 + seed)
        T = gc.max_episode_steps
        actions = rng.integers(0, 5, size=(T, gc.num_agents))
        rec = dict(obs0=np.stack(obs), obs=[], rewards=[], terminated=[], truncated=[], is_active=[], agents_xy=[],
                   targets_xy=[])
        for t in range(T):
            obs, rew, term, trunc, infos = env.step(actions[t].tolist())
            rec["obs"].append(np.stack(obs))
            rec["rewards"].append(rew)
            rec["terminated"].append(term)
            rec["truncated"].append(trunc)
            rec["is_active"].append([i.get("is_active", True) for i in infos])
            rec["agents_xy"].append(grid.get_agents_xy(ignore_borders=True))
            rec["targets_xy"].append(grid.get_targets_xy(ignore_borders=True))
            if all(term) or all(trunc):
                actions = actions[:t + 1]
                break
        name = f"reference_{g['size']}x{g['num_agents']}_{cs}_{ot}_s{seed}.npz"
        np.savez_compressed(os.path.join(out_dir, name), obstacles=obstacles, agents_xy0=agents0, targets_xy0=targets0,
                            actions=actions, obs_radius=r, collision_system=cs, on_target=ot,
                            max_episode_steps=gc.max_episode_steps,
                            **{k: np.asarray(v) for k, v in rec.items()})
        n += 1
    print(f"wrote {n} fixtures to {out_dir}")


if __name__ == "__main__":
    main()
