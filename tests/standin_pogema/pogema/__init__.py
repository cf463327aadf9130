"""STAND-IN for the real `pogema` package -- test infrastructure only, NOT the reference.

The reference (Cognitive-AI-Systems/pogema) is not available in the build container (/root/reference/README.md:3,5), so
tools/gen_golden.py -- the script a maintainer runs against the REAL package to pin parity -- could never be executed
here.  This module gives it something to import: the surface gen_golden.py touches (`GridConfig`, `pogema_v0`,
`env.reset(seed)`, `env.step(list)`, `env.grid.get_obstacles / get_agents_xy / get_targets_xy(ignore_borders=True)`),
implemented on top of the repo's own CPU oracle.  tests/test_golden_pipeline.py puts this directory on PYTHONPATH and
runs the whole generate -> .npz -> load -> compare loop into a temporary directory, so the pipeline is known to work
before the real package arrives.  Fixtures made from this module say nothing about parity with upstream and must never
be written to tests/golden/ (gen_golden.py refuses: `__standin__`).
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import generator_oracle as _G  # noqa: E402
from oracle.pogema_oracle import PogemaOracle as _Oracle  # noqa: E402
from pogema_amd.grid_config import GridConfig  # noqa: E402,F401

__standin__ = True
__version__ = "0.0.0-standin"


class _Grid:
    def __init__(self, env):
        self._env = env

    def get_obstacles(self, ignore_borders=False):
        g = self._env._oracle.grid
        r = g.r
        return (g.obstacles[r:-r, r:-r] if ignore_borders else g.obstacles).copy()

    def get_agents_xy(self, ignore_borders=False):
        g = self._env._oracle.grid
        return [list(p) for p in (g.unpadded_xy(g.positions_xy) if ignore_borders else g.positions_xy)]

    def get_targets_xy(self, ignore_borders=False):
        g = self._env._oracle.grid
        return [list(p) for p in (g.unpadded_xy(g.finishes_xy) if ignore_borders else g.finishes_xy)]

    @property
    def positions(self):  # the padded occupancy array, `Grid.positions` upstream
        return self._env._oracle.grid.positions


class _Env:
    def __init__(self, grid_config):
        self.grid_config = grid_config
        self._oracle = None
        self.grid = _Grid(self)
        self.unwrapped = self

    def reset(self, seed=None, options=None):
        gc = self.grid_config
        seed = gc.seed if seed is None else seed
        h, w = gc.map_shape
        if gc.map is not None and gc.agents_xy is not None:
            obstacles, agents, targets = np.array(gc.map, np.uint8), gc.agents_xy, gc.targets_xy
        else:
            # upstream's generator as recalled, written with numpy itself (the checker of pgx_np_generate)
            obstacles, agents, targets = _G.generate_instance_numpy(seed or 0, h, w, gc.num_agents, gc.density,
                                                                    given_map=gc.map)
        # PGX_STANDIN_SEMANTICS="soft_vertex_rule=all_stay,soft_occupancy=exact,...": a stand-in whose switches are NOT the
        # defaults -- tests/test_golden_pipeline.py checks that tools/pin_reference.sh finds them from the fixtures alone
        sem = dict(item.split("=", 1) for item in os.environ.get("PGX_STANDIN_SEMANTICS", "").split(",") if "=" in item)
        self._oracle = _Oracle(obstacles, agents, targets, obs_radius=gc.obs_radius, collision_system=gc.collision_system,
                               on_target=gc.on_target, max_episode_steps=gc.max_episode_steps, seed=gc.seed or 0, **sem)
        return self._oracle._obs(), [{"is_active": True} for _ in range(gc.num_agents)]

    def step(self, actions):
        return self._oracle.step(actions)


def pogema_v0(grid_config=None):
    return _Env(grid_config if grid_config is not None else GridConfig(num_agents=2))
