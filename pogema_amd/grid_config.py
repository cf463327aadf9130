"""GridConfig -- the reference's configuration surface (upstream `pogema/grid_config.py`).

Same field names, defaults and validation ranges as recalled in SURVEY.md section 8a row A0 (the
mounted reference is a stub, so no file:line can be cited); written natively for pydantic v2.
"""
from __future__ import annotations

from typing import List, Literal, Optional, Union

from pydantic import BaseModel, ConfigDict, field_validator, model_validator

FREE = 0
OBSTACLE = 1
MOVES = [[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]]  # noop, up, down, left, right ; first index = row


def str_map_to_list(str_map: str, free: str = ".", obstacle: str = "#"):
    """Parse a text map: '.' free, '#' obstacle; a lowercase letter marks an agent start and the same
    letter in uppercase its target (both on free cells); '@' a possible start cell, '$' a possible target cell,
    '!' both (recalled from upstream, conf. medium).  Returns (rows, agents_xy, targets_xy) and leaves the possible
    cells in `str_map_to_list.possible` = (possible_agents_xy, possible_targets_xy)."""
    rows, agents, targets = [], {}, {}
    possible_agents, possible_targets = [], []
    for i, line in enumerate(str_map.split()):
        row = []
        for j, ch in enumerate(line):
            if ch == free:
                row.append(FREE)
            elif ch == obstacle:
                row.append(OBSTACLE)
            elif ch.isalpha():
                (targets if ch.isupper() else agents)[ch.lower()] = (i, j)
                row.append(FREE)
            elif ch in "@$!":
                if ch in "@!":
                    possible_agents.append([i, j])
                if ch in "$!":
                    possible_targets.append([i, j])
                row.append(FREE)
            else:
                raise KeyError(f"unsupported symbol {ch!r} at line {i}")
        if row:
            if rows and len(rows[-1]) != len(row):
                raise IndexError(f"map row {i} has width {len(row)}, previous rows have {len(rows[-1])}")
            rows.append(row)
    agents_xy, targets_xy = [], []
    for name in sorted(agents):
        if name not in targets:
            raise KeyError(f"agent {name!r} has no target {name.upper()!r} on the map")
        agents_xy.append(list(agents[name]))
        targets_xy.append(list(targets[name]))
    if set(targets) - set(agents):
        raise KeyError("target without an agent on the map")
    str_map_to_list.possible = (possible_agents, possible_targets)
    return rows, agents_xy, targets_xy


class GridConfig(BaseModel):
    model_config = ConfigDict(validate_assignment=False, extra="forbid")

    FREE: Literal[0] = 0
    OBSTACLE: Literal[1] = 1
    MOVES: list = MOVES

    on_target: Literal["finish", "nothing", "restart"] = "finish"
    seed: Optional[int] = None
    size: int = 8
    density: float = 0.3
    num_agents: int = 1
    obs_radius: int = 5
    agents_xy: Optional[list] = None
    targets_xy: Optional[list] = None
    collision_system: Literal["block_both", "priority", "soft"] = "priority"
    persistent: bool = False
    observation_type: Literal["POMAPF", "MAPF", "default"] = "default"
    map: Optional[Union[List[list], str]] = None
    map_name: Optional[str] = None
    integration: Optional[Literal["SampleFactory", "PyMARL", "rllib", "gymnasium", "PettingZoo"]] = None
    max_episode_steps: int = 64
    auto_reset: Optional[bool] = None
    possible_agents_xy: Optional[list] = None
    possible_targets_xy: Optional[list] = None
    empty_outside: bool = True

    @field_validator("seed")
    @classmethod
    def _seed(cls, v):
        assert v is None or v >= 0, "seed must be positive"
        return v

    @field_validator("size")
    @classmethod
    def _size(cls, v):
        assert 2 <= v <= 1024, "size must be in [2, 1024]"
        return v

    @field_validator("density")
    @classmethod
    def _density(cls, v):
        assert 0.0 <= v <= 1.0, "density must be in [0, 1]"
        return v

    @field_validator("num_agents")
    @classmethod
    def _num_agents(cls, v):
        assert 1 <= v <= 10_000_000, "num_agents must be in [1, 10000000]"
        return v

    @field_validator("obs_radius")
    @classmethod
    def _obs_radius(cls, v):
        assert 1 <= v <= 128, "obs_radius must be in [1, 128]"
        return v

    @model_validator(mode="after")
    def _map(self):
        if self.map is not None:
            if isinstance(self.map, str):
                rows, agents_xy, targets_xy = str_map_to_list(self.map)
                object.__setattr__(self, "map", rows)
                pa, pt = str_map_to_list.possible
                if pa and self.possible_agents_xy is None:
                    object.__setattr__(self, "possible_agents_xy", pa)
                if pt and self.possible_targets_xy is None:
                    object.__setattr__(self, "possible_targets_xy", pt)
                if agents_xy and self.agents_xy is None and self.targets_xy is None:
                    object.__setattr__(self, "agents_xy", agents_xy)
                    object.__setattr__(self, "targets_xy", targets_xy)
                    object.__setattr__(self, "num_agents", len(agents_xy))
            m = self.map
            if not m or any(len(row) != len(m[0]) for row in m):
                raise ValueError("map must be a non-empty rectangular list of rows")
            # `map` overrides `size`; `density` becomes the map's obstacle fraction
            object.__setattr__(self, "size", max(len(m), len(m[0])))
            area = len(m) * len(m[0])
            object.__setattr__(self, "density", sum(1 for row in m for c in row if c != FREE) / area)
        h, w = self.map_shape
        for name in ("agents_xy", "targets_xy"):
            pts = getattr(self, name)
            if pts is not None:
                for p in pts:
                    if len(p) != 2 or not (0 <= p[0] < h and 0 <= p[1] < w):
                        raise IndexError(f"{name} entry {p} is outside the {h}x{w} map")
        if (self.agents_xy is None) != (self.targets_xy is None):
            raise ValueError("agents_xy and targets_xy must be given together")
        if self.agents_xy is not None:
            if len(self.agents_xy) != len(self.targets_xy):
                raise IndexError("agents_xy and targets_xy differ in length")
            object.__setattr__(self, "num_agents", len(self.agents_xy))
        return self

    @property
    def map_shape(self):
        if self.map is not None:
            return len(self.map), len(self.map[0])
        return self.size, self.size


# preset difficulty configs in the style of the reference's registry (`pogema/__init__.py`)
class Easy8x8(GridConfig):
    size: int = 8
    density: float = 0.2
    num_agents: int = 1
    max_episode_steps: int = 64


class Normal8x8(GridConfig):
    size: int = 8
    density: float = 0.3
    num_agents: int = 2
    max_episode_steps: int = 64


class Hard8x8(GridConfig):
    size: int = 8
    density: float = 0.3
    num_agents: int = 4
    max_episode_steps: int = 64


class Easy16x16(GridConfig):
    size: int = 16
    density: float = 0.2
    num_agents: int = 4
    max_episode_steps: int = 128


class Hard16x16(GridConfig):
    size: int = 16
    density: float = 0.3
    num_agents: int = 16
    max_episode_steps: int = 128


class Easy32x32(GridConfig):
    size: int = 32
    density: float = 0.2
    num_agents: int = 16
    max_episode_steps: int = 256


class Hard32x32(GridConfig):
    size: int = 32
    density: float = 0.3
    num_agents: int = 64
    max_episode_steps: int = 256


class Easy64x64(GridConfig):
    size: int = 64
    density: float = 0.2
    num_agents: int = 64
    max_episode_steps: int = 512


class Hard64x64(GridConfig):
    size: int = 64
    density: float = 0.3
    num_agents: int = 256
    max_episode_steps: int = 512
