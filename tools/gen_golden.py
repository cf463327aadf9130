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
Semantics.generator_rng='numpy') against the reference's.  Since round 4 also: the occupancy array itself
(`grid.positions`, padded, after reset and after every step -- settles docs/SPEC.md Q2 by data, not only through the
observation planes) and `infos[0]['metrics']` of the step that ends the episode (Q9: the metric formulas); and one
extra file, reference_probes.json: what the package does with an out-of-range action (Q7) and its GridConfig defaults
(Q8) -- things no rollout of valid actions can show.  tools/pin_reference.sh runs the whole procedure and reports which
position of every semantics switch the fixtures demand.
"""
import json
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


def _unpadded(grid, r, method, attr):
    """Public accessor first (`Grid.get_*_xy(ignore_borders=True)`), else the raw padded attribute minus the border --
    so that a release whose accessor is named or shaped differently still yields fixtures."""
    fn = getattr(grid, method, None)
    if callable(fn):
        try:
            return np.asarray(fn(ignore_borders=True))
        except TypeError:
            pass
    raw = np.asarray(getattr(grid, attr))
    return raw[r:-r, r:-r] if raw.ndim == 2 and attr == "obstacles" else raw - r


def obstacles_of(grid, r):
    return (_unpadded(grid, r, "get_obstacles", "obstacles") != 0).astype(np.uint8)


def agents_of(grid, r):
    return np.asarray(_unpadded(grid, r, "get_agents_xy", "positions_xy"), dtype=np.int32)


def targets_of(grid, r):
    return np.asarray(_unpadded(grid, r, "get_targets_xy", "finishes_xy"), dtype=np.int32)


def grid_of(env):
    """The Grid behind whatever wrappers `pogema_v0` stacked (looked up again after every reset: lifelong environments
    re-create it)."""
    base = env.unwrapped if hasattr(env, "unwrapped") else env
    seen = 0
    while not hasattr(base, "grid") and hasattr(base, "env") and seen < 16:
        base, seen = base.env, seen + 1
    return base.grid


def main():
    ap = argparse.ArgumentParser()
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap.add_argument("--out", default=golden, help="directory for the reference_*.npz fixtures")
    ap.add_argument("--limit", type=int, default=0, help="stop after this many cases (pipeline tests)")
    ap.add_argument("--geoms", default="", help="comma-separated indices of the geometries to run (default: all four; "
                                                "3 = the collision-dense one, which exercises the soft-collision switches)")
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
    if args.geoms:
        geoms = [geoms[int(i)] for i in args.geoms.split(",")]
    n = 0
    failures = []  # a case the package refuses (unknown option in this release, unplaceable instance ...) must not cost the rest
    for g, cs, ot, seed in itertools.product(geoms, ("priority", "block_both", "soft"),
                                             ("finish", "restart", "nothing"), (0, 1, 2)):
        if args.limit and n >= args.limit:
            break
        case = f"{g['size']}x{g['num_agents']}_{cs}_{ot}_s{seed}"
        try:
            n += one_case(GridConfig, pogema_v0, g, cs, ot, seed, out_dir)
        except Exception as exc:  # noqa: BLE001
            failures.append({"case": case, "error": repr(exc)})
            print(f"case {case} FAILED: {exc!r}", file=sys.stderr)
    write_probes(pogema, GridConfig, pogema_v0, out_dir, failures)
    print(f"wrote {n} fixtures to {out_dir}" + (f"; {len(failures)} case(s) failed (reference_probes.json)" if failures else ""))
    if n == 0:
        sys.exit("no fixture could be generated")


def one_case(GridConfig, pogema_v0, g, cs, ot, seed, out_dir):
    gc = GridConfig(seed=seed, collision_system=cs, on_target=ot, max_episode_steps=32, **g)
    env = pogema_v0(gc)
    obs, _ = env.reset(seed=seed)
    grid = grid_of(env)
    r = gc.obs_radius
    obstacles = obstacles_of(grid, r)
    agents0 = agents_of(grid, r)
    targets0 = targets_of(grid, r)
    rng = np.random.default_rng(1000 + seed)
    T = gc.max_episode_steps
    actions = rng.integers(0, 5, size=(T, gc.num_agents))
    greedy = seed == 2  # one seed in three is driven towards the goals (early termination, goal-side conflicts)
    rec = dict(obs0=np.stack(obs), obs=[], rewards=[], terminated=[], truncated=[], is_active=[], agents_xy=[],
               targets_xy=[])
    has_positions = hasattr(grid, "positions")  # the occupancy array (padded), `Grid.positions` upstream
    if has_positions:
        rec["positions0"] = np.asarray(grid.positions, dtype=np.uint8).copy()
        rec["positions"] = []
    metrics = None
    for t in range(T):
        if greedy:
            actions[t] = greedy_actions(obstacles, agents_of(grid, r), targets_of(grid, r), rng)
        obs, rew, term, trunc, infos = env.step(actions[t].tolist())
        rec["obs"].append(np.stack(obs))
        rec["rewards"].append(rew)
        rec["terminated"].append(term)
        rec["truncated"].append(trunc)
        rec["is_active"].append([i.get("is_active", True) for i in infos])
        rec["agents_xy"].append(agents_of(grid, r))
        rec["targets_xy"].append(targets_of(grid, r))
        if has_positions:
            rec["positions"].append(np.asarray(grid.positions, dtype=np.uint8).copy())
        if isinstance(infos[0], dict) and isinstance(infos[0].get("metrics"), dict):
            metrics = (t, infos[0]["metrics"])
        if all(term) or all(trunc):
            actions = actions[:t + 1]
            break
    extra = {}
    if metrics is not None:  # names as one '|'-joined string (no pickles in the fixtures), values in the same order
        names = sorted(metrics[1])
        extra = dict(metrics_step=metrics[0], metrics_names="|".join(names),
                     metrics_values=np.asarray([float(metrics[1][k]) for k in names], dtype=np.float64))
    name = f"reference_{g['size']}x{g['num_agents']}_{cs}_{ot}_s{seed}.npz"
    np.savez_compressed(os.path.join(out_dir, name), obstacles=obstacles, agents_xy0=agents0, targets_xy0=targets0,
                        actions=actions, obs_radius=r, collision_system=cs, on_target=ot,
                        max_episode_steps=gc.max_episode_steps, grid_seed=seed, density=gc.density, **extra,
                        **{k: np.asarray(v) for k, v in rec.items()})
    return 1


def write_probes(pogema, GridConfig, pogema_v0, out_dir, failures):
    # ---- probes: behaviour no rollout of valid actions shows -------------------------------------------------
    probes = {"package": getattr(pogema, "__name__", "pogema"), "version": str(getattr(pogema, "__version__", "?")),
              "standin": bool(getattr(pogema, "__standin__", False)), "failed_cases": failures}
    try:  # docs/SPEC.md Q7: an action outside 0..4
        env = pogema_v0(GridConfig(seed=0, size=8, num_agents=2, obs_radius=2, density=0.1))
        env.reset(seed=0)
        g0 = grid_of(env)
        # Q11: the declared spaces (metadata only: recalled as Box(-1.0, 1.0, (3, W, W)) for the default observation type,
        # SURVEY says Box(0, 1, ...); the mirror declares 0..1)
        probes["observation_space"] = repr(getattr(env, "observation_space", None))
        probes["action_space"] = repr(getattr(env, "action_space", None))
        before = [list(map(int, p)) for p in agents_of(g0, 2)]
        try:
            env.step([7, 0])
            after = [list(map(int, p)) for p in agents_of(g0, 2)]
            probes["bad_action"] = "noop" if after == before else f"moved {before} -> {after}"
        except Exception as exc:  # noqa: BLE001
            probes["bad_action"] = f"raises {type(exc).__name__}"
    except Exception as exc:  # noqa: BLE001
        probes["bad_action"] = f"probe failed: {exc!r}"
    try:  # Q8: the defaults of GridConfig
        d = GridConfig()
        probes["grid_config_defaults"] = {k: (getattr(d, k) if isinstance(getattr(d, k, None), (int, float, str, bool, type(None)))
                                              else repr(getattr(d, k, None)))
                                          for k in ("on_target", "seed", "size", "density", "num_agents", "obs_radius",
                                                    "collision_system", "observation_type", "max_episode_steps", "persistent",
                                                    "empty_outside", "auto_reset", "integration")}
    except Exception as exc:  # noqa: BLE001
        probes["grid_config_defaults"] = f"probe failed: {exc!r}"
    with open(os.path.join(out_dir, "reference_probes.json"), "w") as f:
        json.dump(probes, f, indent=1, default=str)


if __name__ == "__main__":
    main()
