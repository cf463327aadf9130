#!/usr/bin/env python3
"""Reference fixtures from upstream's grid layer ALONE -- for a container that has the reference's SOURCE but not its
dependencies.

`import pogema` executes `pogema/__init__.py`, which needs `gymnasium`; `pogema/grid_config.py`, `pogema/generator.py` and
`pogema/grid.py` need only numpy + pydantic.  This script imports exactly those three modules from the source tree under
`--ref` WITHOUT running the package's `__init__` (a bare module object named `pogema` whose `__path__` is the source
directory stands in for it) and drives upstream's own `Grid`:

    python tools/gen_golden_grid.py --ref /root/reference [--out tests/golden] [--limit N]

Per case (geometry x seed) it records what `Grid(GridConfig(seed=..., ...))` built -- the unpadded map, the PADDED obstacle
array with its border ring, starts and targets (the numpy-stream instance generator, docs/SPEC.md Q6) -- and
then a scripted episode of `grid.move(agent, action)` calls in agent-index order, which is literally what
`collision_system='priority'` does (A2/A3), with every agent's `get_obstacles_for_agent / get_positions /
get_square_target` planes after every step (A9-A11) and the occupancy array `grid.positions` itself.  What it cannot
record is anything `pogema/envs.py` adds (rewards, done flags, `block_both` / `soft`, lifelong targets): that needs
tools/gen_golden.py and an importable package.  Output: reference_grid_*.npz, checked by tests/test_golden_reference.py
(`compare_grid_fixture`: oracle on the CPU, engine on the GPU, the numpy-stream generator against the recorded instance).
The rehearsal in tests/test_golden_pipeline.py runs it against a stand-in source tree whose `__init__.py` refuses to import.
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np


def load_grid_layer(ref):
    """-> (GridConfig, Grid) of the source tree under `ref`, without executing pogema/__init__.py."""
    src = os.path.join(os.path.abspath(ref), "pogema")
    for name in ("grid_config.py", "grid.py"):
        if not os.path.exists(os.path.join(src, name)):
            sys.exit(f"{src}/{name} not found: --ref must point at a checkout of the reference (the directory that holds pogema/)")
    for stale in [m for m in sys.modules if m == "pogema" or m.startswith("pogema.")]:
        del sys.modules[stale]
    pkg = types.ModuleType("pogema")
    pkg.__path__ = [src]          # a namespace for `from pogema.generator import ...`; __init__.py is never run
    pkg.__grid_layer_only__ = True
    sys.modules["pogema"] = pkg
    try:
        gc = importlib.import_module("pogema.grid_config")
        grid = importlib.import_module("pogema.grid")
    except Exception as exc:  # noqa: BLE001
        sys.exit(f"the grid layer of {src} is not importable on its own ({exc!r})")
    return gc.GridConfig, grid.Grid, bool(getattr(grid, "__standin__", False))


def main():
    ap = argparse.ArgumentParser()
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=golden)
    ap.add_argument("--limit", type=int, default=0)
    args = ap.parse_args()
    GridConfig, Grid, standin = load_grid_layer(args.ref)
    out_dir = os.path.abspath(args.out)
    if standin and os.path.realpath(out_dir) == os.path.realpath(golden):
        sys.exit("the source tree under --ref is the repo's STAND-IN: refusing to write its output into tests/golden/")
    os.makedirs(out_dir, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gen_golden import agents_of, obstacles_of, targets_of  # the same accessor fallbacks
    geoms = [dict(size=8, num_agents=2, obs_radius=3, density=0.3), dict(size=16, num_agents=8, obs_radius=5, density=0.3),
             dict(size=32, num_agents=16, obs_radius=5, density=0.3), dict(size=12, num_agents=40, obs_radius=2, density=0.1)]
    n, failures = 0, []
    for g in geoms:
        for seed in (0, 1, 2, 3):
            if args.limit and n >= args.limit:
                break
            try:
                gc = GridConfig(seed=seed, **g)
                try:
                    grid = Grid(grid_config=gc)
                except TypeError:
                    grid = Grid(gc)
                r, A = gc.obs_radius, gc.num_agents
                rec = dict(obstacles=obstacles_of(grid, r), padded_obstacles=(np.asarray(grid.obstacles) != 0).astype(np.uint8),
                           agents_xy0=agents_of(grid, r), targets_xy0=targets_of(grid, r))

                def planes():
                    return np.stack([np.stack([np.asarray(grid.get_obstacles_for_agent(i), dtype=np.float32),
                                               np.asarray(grid.get_positions(i), dtype=np.float32),
                                               np.asarray(grid.get_square_target(i), dtype=np.float32)]) for i in range(A)])

                rec["obs0"] = planes()
                has_positions = hasattr(grid, "positions")
                if has_positions:
                    rec["positions0"] = (np.asarray(grid.positions) != 0).astype(np.uint8)
                rng = np.random.default_rng(2000 + seed)
                T = 24
                actions = rng.integers(0, 5, size=(T, A))
                obs, xy, occ = [], [], []
                for t in range(T):
                    for i in range(A):            # `Pogema.move_agents`, collision_system='priority': index order, each
                        grid.move(i, int(actions[t, i]))  # move sees the earlier ones
                    obs.append(planes())
                    xy.append(agents_of(grid, r))
                    if has_positions:
                        occ.append((np.asarray(grid.positions) != 0).astype(np.uint8))
                    if all(tuple(a) == tuple(b) for a, b in zip(xy[-1], rec["targets_xy0"])):
                        actions = actions[:t + 1]  # everybody on its goal at once: an episode of envs.py would end here
                        break
                rec.update(actions=actions, obs=np.stack(obs), agents_xy=np.stack(xy), obs_radius=r, grid_seed=seed, density=gc.density)
                if has_positions:
                    rec["positions"] = np.stack(occ)
                np.savez_compressed(os.path.join(out_dir, f"reference_grid_{g['size']}x{A}_s{seed}.npz"), **rec)
                n += 1
            except Exception as exc:  # noqa: BLE001
                failures.append(f"{g['size']}x{g['num_agents']}_s{seed}: {exc!r}")
                print(f"case {g['size']}x{g['num_agents']}_s{seed} FAILED: {exc!r}", file=sys.stderr)
    print(f"wrote {n} grid-layer fixtures to {out_dir}" + (f"; {len(failures)} failed" if failures else ""))
    if n == 0:
        sys.exit("no fixture could be generated")


if __name__ == "__main__":
    main()
