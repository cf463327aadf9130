"""STAND-IN (test infrastructure) for upstream `pogema/grid.py`: `Grid(grid_config)` with upstream's method names, built on
the repo's oracle.  Says nothing about parity with upstream; tools/gen_golden_grid.py refuses to write its output into
tests/golden/ (`__standin__`)."""
import numpy as np

from pogema.generator import generate_instance_numpy   # absolute imports through the bare `pogema` namespace, as upstream
from pogema.grid_config import GridConfig  # noqa: F401

from oracle.pogema_oracle import Grid as _OracleGrid

__standin__ = True


class Grid(_OracleGrid):
    def __init__(self, grid_config, add_artificial_border=True, num_retries=10):
        gc = grid_config
        h, w = gc.map_shape
        obstacles, agents, targets = generate_instance_numpy(gc.seed or 0, h, w, gc.num_agents, gc.density, given_map=gc.map)
        super().__init__(obstacles, agents, targets, gc.obs_radius)
        self.config = gc

    def get_obstacles(self, ignore_borders=False):
        r = self.r
        return (self.obstacles[r:-r, r:-r] if ignore_borders else self.obstacles).copy()

    def get_agents_xy(self, only_active=False, ignore_borders=False):
        return [list(p) for p in (self.unpadded_xy(self.positions_xy) if ignore_borders else self.positions_xy)]

    def get_targets_xy(self, only_active=False, ignore_borders=False):
        return [list(p) for p in (self.unpadded_xy(self.finishes_xy) if ignore_borders else self.finishes_xy)]
