#!/usr/bin/env python3
"""Generate REAL reference fixtures -- run this where `import pogema` works.

In this build container it cannot run: /root/reference holds only README.md and neither `pogema` nor
`gymnasium` is installed.  It is committed so that a maintainer with the real package can convert
"parity unpinned" into a pinned contract:

    pip install pogema            # or: PYTHONPATH=/path/to/Cognitive-AI-Systems/pogema
    python tools/gen_golden.py    # writes tests/golden/reference_*.npz      [--out DIR] [--limit N]

The pipeline itself (this script -> .npz -> tests/test_golden_reference.py's loader -> comparison) is exercised in CI
against a stand-in `pogema` built from the repo's oracle (tests/standin_pogema, tests/test_golden_pipeline.py), always
into a temporary directory: this script refuses to write stand-in output into tests/golden/.

tests/test_golden_reference.py then checks the oracle (CPU) and the HIP engine (GPU) against every
fixture found.  The reference's Python never travels: only the .npz vectors (inputs + expected
outputs) are committed.

What is recorded per case: the initial state actually used by the reference (obstacles, agents_xy,
targets_xy read back from its Grid, unpadded), the action stream, and per step agents_xy, targets_xy,
is_active, rewards, terminated, truncated and the full float32 observations; plus the GridConfig numbers the random
instance came from (grid_seed, density), which pin the numpy-stream instance generator (pgx_np_generate,
Semantics.generator_rng='numpy') against the reference's.
"""
import argparse
import itertools
import os
import sys

import numpy as np


MOVES = ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1))


def greedy_actions(obstacles, agents_xy, targets_xy, rng):
    """A goal-seeking policy computed from the PUBLIC accessors only (so that some recorded episodes end before the time
    limit and conflicts cluster around goals): step along the larger coordinate difference when that cell is free,
    else along the other one, else stay; 25 % random moves."""
    h, w = obstacles.shape
    acts = []
    for (x, y), (tx, ty) in zip(agents_xy, targets_xy):
        if rng.random() < 0.25:
            acts.append(int(rng.integers(0, 5)))
            continue
        dx, dy = tx - x, ty - y
        prefs = []
        if dx != 0:
            prefs.append(1 if dx < 0 else 2)
        if dy != 0:
            prefs.append(3 if dy < 0 else 4)
        if abs(dy) > abs(dx):
            prefs.reverse()
        choice = 0
        for a in prefs:
            nx, ny = x + MOVES[a][0], y + MOVES[a][1]
            if 0 <= nx < h and 0 <= ny < w and obstacles[nx, ny] == 0:
                choice = a
                break
        acts.append(choice)
    return acts


def main():
    ap = argparse.ArgumentParser()
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap.add_argument("--out", default=golden, help="directory for the reference_*.npz fixtures")
    ap.add_argument("--limit", type=int, default=0, help="stop after this many cases (pipeline tests)")
    args = ap.parse_args()
    try:
        import pogema
        from pogema import GridConfig, pogema_v0
    except Exception as exc:  # pragma: no cover - depends on the environment
        sys.exit(f"pogema is not importable here ({exc!r}); nothing generated")
    out_dir = os.path.abspath(args.out)
    if getattr(pogema, "__standin__", False) and os.path.realpath(out_dir) == os.path.realpath(golden):
        sys.exit("the importable `pogema` is the repo's STAND-IN (tests/standin_pogema): refusing to write its output "
                 "into tests/golden/ -- fixtures there must come from the real package")
    os.makedirs(out_dir, exist_ok=True)
    geoms = [dict(size=8, num_agents=2, obs_radius=3, density=0.3),      # BASELINE.json configs[0]
             dict(size=16, num_agents=8, obs_radius=5, density=0.3),
             dict(size=32, num_agents=16, obs_radius=5, density=0.3),
             dict(size=12, num_agents=40, obs_radius=2, density=0.1)]    # collision-dense
    n = 0
    for g, cs, ot, seed in itertools.product(geoms, ("priority", "block_both", "soft"),
                                             ("finish", "restart", "nothing"), (0, 1, 2)):
        if args.limit and n >= args.limit:
            break
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
        greedy = seed == 2  # one seed in three is driven towards the goals (early termination, goal-side conflicts)
        rec = dict(obs0=np.stack(obs), obs=[], rewards=[], terminated=[], truncated=[], is_active=[], agents_xy=[],
                   targets_xy=[])
        for t in range(T):
            if greedy:
                actions[t] = greedy_actions(obstacles, grid.get_agents_xy(ignore_borders=True),
                                            grid.get_targets_xy(ignore_borders=True), rng)
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
                            max_episode_steps=gc.max_episode_steps, grid_seed=seed, density=gc.density,
                            **{k: np.asarray(v) for k, v in rec.items()})
        n += 1
    print(f"wrote {n} fixtures to {out_dir}")


if __name__ == "__main__":
    main()
