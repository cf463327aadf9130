"""CPU ORACLE (test infrastructure only) -- literal pure-Python restatement of POGEMA's step path.

    *** PARITY UNPINNED ***
    /root/reference holds only a 5-line README (README.md:3,5 say the code lives in another
    repository), `pogema`/`gymnasium` are not importable here, and no reference test or golden
    vector exists in this container.  Everything below follows SURVEY.md section 8(a) rows A0..A13
    (the written SPEC) plus the builder's recollection of upstream `pogema/grid.py`,
    `pogema/envs.py` and `pogema/wrappers/multi_time_limit.py` (1.3-era).  No reference line numbers
    are cited because none could be opened.  Open parity questions are listed in DESIGN.md.

This module is imported ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (pogema_amd/) never imports it.

Data structures deliberately mirror the reference's (dicts keyed by (x, y) tuples, python lists,
recursion) so that order-dependent behaviour (agent-index order, reverse-index revert order,
list insertion order) is reproduced by construction rather than re-derived.

Coordinates: `x` is the ROW index, `y` the COLUMN index (MOVES first index is row).
All coordinates held by `Grid` are PADDED coordinates (shifted by +obs_radius), as upstream does
after `add_artificial_border`.
"""
from __future__ import annotations

import numpy as np

FREE = 0
OBSTACLE = 1
# noop, up, down, left, right ; first index is the row.
MOVES = ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1))

COLLISION_SYSTEMS = ("priority", "block_both", "soft")
ON_TARGET = ("finish", "restart", "nothing")
# Switches for the low-confidence recollections (docs/SPEC.md Q1 / Q4 / Q7); the first value is the recalled default.
SOFT_VERTEX_RULES = ("lowest_index", "all_stay")
COOP_REWARDS = ("all_solved", "per_agent")
BAD_ACTIONS = ("noop", "flag")

_MASK32 = 0xFFFFFFFF
TAG_OUTSIDE = 0x4F55545300000000  # 'OUTS'
_MASK64 = 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------------------------
# Lifelong target RNG.  The reference draws from per-agent numpy PCG64 generators
# (SURVEY A7); that stream cannot be reproduced without the source, so the build defines its own
# counter-based generator.  It is part of the build's SPEC (docs/SPEC.md S5, lifelong stream) and is
# shared verbatim by oracle/pogema_oracle.c and the HIP kernel.
# ----------------------------------------------------------------------------------------------
def splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


def lifelong_draw(seed: int, env_index: int, agent: int, counter: int, n: int) -> int:
    """Uniform index in [0, n) for the `counter`-th target of `agent` in global env `env_index`."""
    h = splitmix64(seed & _MASK64)
    h = splitmix64(h ^ (env_index & _MASK64))
    h = splitmix64(h ^ ((agent & _MASK32) << 32 | (counter & _MASK32)))
    return ((h >> 32) * n) >> 32


def label_components(obstacles: np.ndarray):
    """4-connected components of FREE cells, ids in row-major order of first cell (0-based);
    obstacle cells get -1.  Returns (labels, list_of_point_lists) with points row-major."""
    h, w = obstacles.shape
    labels = -np.ones((h, w), dtype=np.int64)
    comps = []
    for sx in range(h):
        for sy in range(w):
            if obstacles[sx, sy] != FREE or labels[sx, sy] >= 0:
                continue
            cid = len(comps)
            labels[sx, sy] = cid
            stack = [(sx, sy)]
            while stack:
                x, y = stack.pop()
                for dx, dy in MOVES[1:]:
                    nx, ny = x + dx, y + dy
                    if 0 <= nx < h and 0 <= ny < w and obstacles[nx, ny] == FREE and labels[nx, ny] < 0:
                        labels[nx, ny] = cid
                        stack.append((nx, ny))
            comps.append(None)
    pts = [[] for _ in comps]
    for x in range(h):
        for y in range(w):
            if labels[x, y] >= 0:
                pts[labels[x, y]].append((x, y))
    return labels, pts


class Grid:
    """SURVEY A1/A2/A9/A10/A11 (upstream `pogema/grid.py: Grid`)."""

    def __init__(self, obstacles, agents_xy, targets_xy, obs_radius, empty_outside=True, outside=None):
        obstacles = np.asarray(obstacles)
        assert obstacles.ndim == 2
        self.r = int(obs_radius)
        self.map_h, self.map_w = obstacles.shape
        self.num_agents = len(agents_xy)
        assert len(targets_xy) == self.num_agents
        self._raw_obstacles = obstacles.astype(np.int32)
        # --- add_artificial_border (A1): pad by r, wall ring at offset r-1, outside FREE -------
        r = self.r
        filled = np.zeros((self.map_h + 2 * r, self.map_w + 2 * r), dtype=np.int32)
        height, width = filled.shape
        if not empty_outside:
            # upstream: `rnd.binomial(1, density, padded shape)` from the grid's numpy generator (not reproducible);
            # the build's own stream (docs/SPEC.md S1): a pure function of (seed, global env, generation, cell)
            seed, env_index, epoch, density = outside
            thr = int(np.floor(float(np.float32(density)) * 16777216.0 + 0.5))
            h = splitmix64(splitmix64(splitmix64(seed & _MASK64) ^ (env_index & _MASK64)) ^ (TAG_OUTSIDE | (epoch & _MASK32)))
            for px in range(height):
                for py in range(width):
                    if (splitmix64(h ^ (px * width + py)) >> 40) < thr:
                        filled[px, py] = OBSTACLE
            filled[r - 1:height - r + 1, r - 1:width - r + 1] = FREE
        filled[r - 1, r - 1:width - r + 1] = OBSTACLE
        filled[r - 1:height - r + 1, r - 1] = OBSTACLE
        filled[height - r, r - 1:width - r + 1] = OBSTACLE
        filled[r - 1:height - r + 1, width - r] = OBSTACLE
        filled[r:height - r, r:width - r] = self._raw_obstacles
        self.obstacles = filled
        self.positions_xy = [(int(x) + r, int(y) + r) for x, y in agents_xy]
        self.finishes_xy = [(int(x) + r, int(y) + r) for x, y in targets_xy]
        for x, y in self.positions_xy + self.finishes_xy:
            if self.obstacles[x, y] != FREE:
                raise KeyError("agent or target placed on an obstacle")
        if len(set(self.positions_xy)) != self.num_agents:
            raise KeyError("two agents share a start cell")
        # occupancy array (A1): 1 where an active, non-hidden agent stands
        self.positions = np.zeros(filled.shape, dtype=np.int32)
        for x, y in self.positions_xy:
            self.positions[x, y] = OBSTACLE
        self.is_active = {i: True for i in range(self.num_agents)}

    # -- A2 ------------------------------------------------------------------------------------
    def move(self, agent_id, action):
        x, y = self.positions_xy[agent_id]
        dx, dy = MOVES[action]
        if self.obstacles[x + dx, y + dy] == FREE:
            if self.positions[x + dx, y + dy] == FREE:
                self.positions[x, y] = FREE
                x += dx
                y += dy
                self.positions[x, y] = OBSTACLE
        self.positions_xy[agent_id] = (x, y)

    def has_obstacle(self, x, y):
        return self.obstacles[x, y] == OBSTACLE

    def on_goal(self, agent_id):
        return self.positions_xy[agent_id] == self.finishes_xy[agent_id]

    def hide_agent(self, agent_id):
        if not self.is_active[agent_id]:
            return False
        self.is_active[agent_id] = False
        self.positions[self.positions_xy[agent_id]] = FREE
        return True

    # -- A9 / A10 / A11 --------------------------------------------------------------------------
    def get_obstacles_for_agent(self, agent_id):
        x, y = self.positions_xy[agent_id]
        r = self.r
        return self.obstacles[x - r:x + r + 1, y - r:y + r + 1].astype(np.float32)

    def get_positions(self, agent_id):
        x, y = self.positions_xy[agent_id]
        r = self.r
        return self.positions[x - r:x + r + 1, y - r:y + r + 1].astype(np.float32)

    def get_square_target(self, agent_id):
        r = self.r
        full = 2 * r + 1
        result = np.zeros((full, full), dtype=np.float32)
        x, y = self.positions_xy[agent_id]
        fx, fy = self.finishes_xy[agent_id]
        dx, dy = x - fx, y - fy
        dx = min(dx, r) if dx >= 0 else max(dx, -r)
        dy = min(dy, r) if dy >= 0 else max(dy, -r)
        result[r - dx, r - dy] = 1.0
        return result

    def unpadded_xy(self, padded):
        return [(x - self.r, y - self.r) for x, y in padded]


class PogemaOracle:
    """SURVEY A3..A8, A12, A13 (upstream `pogema/envs.py`: Pogema / PogemaLifeLong /
    PogemaCoopFinish, wrapped by `MultiTimeLimit` and optionally the auto-reset wrapper).

    One instance == one environment; batching is a python loop in the tests.
    """

    def __init__(self, obstacles, agents_xy, targets_xy, obs_radius=5, collision_system="priority",
                 on_target="finish", max_episode_steps=64, auto_reset=False, seed=0, env_index=0,
                 empty_outside=True, outside_density=0.0, epoch=0, soft_vertex_rule="lowest_index", soft_occupancy="index_order",
                 coop_reward="all_solved", bad_action="noop", lifelong_rng="build"):
        assert collision_system in COLLISION_SYSTEMS and on_target in ON_TARGET
        assert soft_vertex_rule in SOFT_VERTEX_RULES and coop_reward in COOP_REWARDS and bad_action in BAD_ACTIONS
        self.soft_vertex_rule, self.coop_reward, self.bad_action = soft_vertex_rule, coop_reward, bad_action
        assert soft_occupancy in ("exact", "index_order")
        self.soft_occupancy = soft_occupancy
        assert lifelong_rng in ("build", "numpy")
        self.lifelong_rng = lifelong_rng
        self._init_args = (np.array(obstacles, copy=True), [tuple(map(int, p)) for p in agents_xy],
                           [tuple(map(int, p)) for p in targets_xy])
        self.obs_radius = int(obs_radius)
        self.collision_system = collision_system
        self.on_target = on_target
        self.max_episode_steps = int(max_episode_steps)
        self.auto_reset = bool(auto_reset)
        self.seed = int(seed)
        self.env_index = int(env_index)
        self.empty_outside = bool(empty_outside)
        self._outside = (self.seed, self.env_index, int(epoch), float(outside_density))
        self.num_agents = len(agents_xy)
        self.reset()

    # ------------------------------------------------------------------------------------------
    def _reset_metrics(self):
        # accumulators of the metrics wrappers (upstream `pogema/wrappers/metrics.py`):
        # number of agents that reached their goal, sum and max of their solve steps, goals in lifelong mode
        self._m_solved = 0
        self._m_sum = 0
        self._m_max = 0
        self._m_goals = 0

    def reset(self):
        self._reset_metrics()
        obstacles, agents_xy, targets_xy = self._init_args
        self.grid = Grid(obstacles, agents_xy, targets_xy, self.obs_radius, empty_outside=self.empty_outside,
                         outside=self._outside)
        self._elapsed_steps = 0
        if self.on_target == "restart":
            labels, pts = label_components(np.asarray(obstacles))
            self._labels, self._comp_points = labels, pts
            # counters survive auto-reset on purpose: the reference re-seeds its generators at
            # reset; the build keeps a monotone per-agent counter instead.
            if not hasattr(self, "_target_counter"):
                self._target_counter = [0] * self.num_agents
            if self.lifelong_rng == "numpy":
                # upstream PogemaLifeLong._initialize_grid (recalled, conf. medium), with numpy itself:
                main_rng = np.random.default_rng(self.seed + self.env_index)
                seeds = main_rng.integers(np.iinfo(np.int32).max, size=self.num_agents)
                self._random_generators = [np.random.default_rng(int(s)) for s in seeds]
        return self._obs()

    # -- A12 -----------------------------------------------------------------------------------
    def _obs(self):
        g = self.grid
        return [np.concatenate([g.get_obstacles_for_agent(i)[None], g.get_positions(i)[None],
                                g.get_square_target(i)[None]]) for i in range(self.num_agents)]

    # -- A5 helper -------------------------------------------------------------------------------
    def _revert_action(self, agent_idx, used_cells, cell, actions):
        actions[agent_idx] = 0
        used_cells[cell].remove(agent_idx)
        new_cell = self.grid.positions_xy[agent_idx]
        if new_cell in used_cells and len(used_cells[new_cell]) > 0:
            used_cells[new_cell].append(agent_idx)
            return self._revert_action(used_cells[new_cell][0], used_cells, new_cell, actions)
        used_cells.setdefault(new_cell, []).append(agent_idx)
        return actions, used_cells

    # -- A3 / A4 / A5 ------------------------------------------------------------------------------
    def move_agents(self, actions):
        g = self.grid
        n = self.num_agents
        if self.collision_system == "priority":
            for i in range(n):
                if g.is_active[i]:
                    g.move(i, actions[i])
        elif self.collision_system == "block_both":
            used_cells = {}
            agents_xy = list(g.positions_xy)
            for i, (x, y) in enumerate(agents_xy):
                if g.is_active[i]:
                    dx, dy = MOVES[actions[i]]
                    used_cells[x + dx, y + dy] = "blocked" if (x + dx, y + dy) in used_cells else "visited"
                    used_cells[x, y] = "blocked"
            for i in range(n):
                if g.is_active[i]:
                    x, y = agents_xy[i]
                    dx, dy = MOVES[actions[i]]
                    if used_cells.get((x + dx, y + dy), None) != "blocked":
                        g.move(i, actions[i])
        elif self.soft_vertex_rule == "all_stay":
            # docs/SPEC.md Q1 alternative -- the textbook MAPF rule, stated as a synchronous (Jacobi) fixed point:
            # in every round ALL movers whose destination is an obstacle, is claimed by anybody else (a non-mover
            # claims the cell it stands on) or lies across a swapped edge are reverted to noop at once; repeat
            # until nothing changes; then the surviving moves are applied.
            actions = list(actions)
            agents_xy = list(g.positions_xy)
            standing = {agents_xy[i]: i for i in range(n) if g.is_active[i]}
            changed = True
            while changed:
                changed = False
                dest = {}
                claims = {}
                for i, (x, y) in enumerate(agents_xy):
                    if g.is_active[i]:
                        dx, dy = MOVES[actions[i]]
                        dest[i] = (x + dx, y + dy)
                        claims.setdefault(dest[i], []).append(i)
                revert = []
                for i in range(n):
                    if g.is_active[i] and actions[i] != 0:
                        d = dest[i]
                        o = standing.get(d)
                        swap = o is not None and actions[o] != 0 and dest[o] == agents_xy[i]
                        if g.has_obstacle(*d) or len(claims[d]) > 1 or swap:
                            revert.append(i)
                for i in revert:
                    actions[i] = 0
                    changed = True
            self._apply_soft_moves(actions)
        else:  # soft, recalled literal algorithm (lowest index wins a contested cell)
            actions = list(actions)
            used_cells = {}
            used_edges = {}
            agents_xy = list(g.positions_xy)
            for i, (x, y) in enumerate(agents_xy):
                if g.is_active[i]:
                    dx, dy = MOVES[actions[i]]
                    used_cells.setdefault((x + dx, y + dy), []).append(i)
                    used_edges[x, y, x + dx, y + dy] = [i]
                    if dx != 0 or dy != 0:
                        used_edges.setdefault((x + dx, y + dy, x, y), []).append(i)
            for i, (x, y) in enumerate(agents_xy):
                if g.is_active[i]:
                    dx, dy = MOVES[actions[i]]
                    if len(used_edges[x, y, x + dx, y + dy]) > 1:
                        used_cells[x + dx, y + dy].remove(i)
                        used_cells.setdefault((x, y), []).append(i)
                        actions[i] = 0
            for i in reversed(range(n)):
                x, y = agents_xy[i]
                if g.is_active[i]:
                    dx, dy = MOVES[actions[i]]
                    if len(used_cells[x + dx, y + dy]) > 1 or g.has_obstacle(x + dx, y + dy):
                        self._revert_action(i, used_cells, (x + dx, y + dy), actions)
            self._apply_soft_moves(actions)

    def _apply_soft_moves(self, actions):
        """`move_without_checks` for every active agent (the surviving moves are mutually compatible).  docs/SPEC.md Q2:
        'index_order' (default, the recalled literal): the per-agent loop -- clear the old cell, set the new one, agent by
        agent in index order -- in which an agent that enters the cell a HIGHER-index agent is leaving has its new cell
        cleared again by that agent's turn: it stands there but is missing from the occupancy array (and from everybody's
        `agents` plane) until its next turn in a later step re-sets it; 'exact' (the alternative): the occupancy array
        afterwards is exactly the set of active agents' cells."""
        g = self.grid
        n = self.num_agents
        if self.soft_occupancy == "index_order":
            for i in range(n):
                if g.is_active[i]:
                    x, y = g.positions_xy[i]
                    dx, dy = MOVES[actions[i]]
                    g.positions[x, y] = FREE
                    g.positions[x + dx, y + dy] = OBSTACLE
                    g.positions_xy[i] = (x + dx, y + dy)
            return
        for i in range(n):
            if g.is_active[i]:
                x, y = g.positions_xy[i]
                g.positions[x, y] = FREE
        for i in range(n):
            if g.is_active[i]:
                x, y = g.positions_xy[i]
                dx, dy = MOVES[actions[i]]
                g.positions_xy[i] = (x + dx, y + dy)
                g.positions[x + dx, y + dy] = OBSTACLE

    # -- A6 / A7 / A8 + A13 + auto-reset ---------------------------------------------------------
    def step(self, actions):
        actions = [int(a) for a in actions]
        assert len(actions) == self.num_agents
        g = self.grid
        n = self.num_agents
        # docs/SPEC.md Q7: the reference indexes MOVES[action] for ACTIVE agents only.  'noop': out-of-range actions
        # do nothing; 'flag': they raise the reference's IndexError (negative values included -- Python's
        # wrap-around of MOVES[-1] is deliberately not reproduced).
        for i, a in enumerate(actions):
            if not 0 <= a <= 4:
                if self.bad_action == "flag" and g.is_active[i]:
                    raise IndexError(f"action {a} of agent {i} is outside 0..4")
                actions[i] = 0
        self.move_agents(actions)
        # `was_on_goal` of the reference: on goal and still active right after the moves
        was_on_goal = [bool(g.on_goal(i) and g.is_active[i]) for i in range(n)]
        if self.on_target == "finish":
            rewards, terminated = [], []
            for i in range(n):
                on_goal = g.on_goal(i)
                rewards.append(1.0 if (on_goal and g.is_active[i]) else 0.0)
                terminated.append(bool(on_goal))
            for i in range(n):
                if g.on_goal(i):
                    g.hide_agent(i)
                    g.is_active[i] = False
        elif self.on_target == "restart":
            rewards = []
            for i in range(n):
                on_goal = g.on_goal(i)
                rewards.append(1.0 if (on_goal and g.is_active[i]) else 0.0)
                if on_goal:
                    g.finishes_xy[i] = self._generate_new_target(i)
            terminated = [False] * n
        else:  # nothing (cooperative finish)
            solved = all(g.on_goal(i) and g.is_active[i] for i in range(n))
            if self.coop_reward == "all_solved":
                rewards = [1.0 if solved else 0.0] * n
            else:  # docs/SPEC.md Q4 alternative: each agent is paid for standing on its own goal
                rewards = [1.0 if (g.on_goal(i) and g.is_active[i]) else 0.0 for i in range(n)]
            terminated = [bool(solved)] * n
        infos = [{"is_active": bool(g.is_active[i])} for i in range(n)]
        truncated = [False] * n
        # A13 MultiTimeLimit
        self._elapsed_steps += 1
        if self._elapsed_steps >= self.max_episode_steps:
            truncated = [True] * n
        finished = all(terminated) or all(truncated)
        metrics = self._compute_metrics(self._elapsed_steps - 1, was_on_goal, finished)
        if metrics is not None:
            infos[0]["metrics"] = metrics
        obs = self._obs()
        if self.auto_reset and finished:
            obs = self.reset()
        elif finished:
            self._reset_metrics()
        return obs, rewards, terminated, truncated, infos

    def _compute_metrics(self, step, was_on_goal, finished):
        """Metric wrappers restated (recollection of upstream `pogema/wrappers/metrics.py`, conf. med --
        docs/SPEC.md Q9): ISR / CSR / ep_length / SoC / makespan for disappearing agents,
        their 'NonDisappear' forms for on_target='nothing', avg_throughput for lifelong."""
        n = self.num_agents
        if self.on_target == "finish":
            for hit in was_on_goal:
                if hit:
                    self._m_solved += 1
                    self._m_sum += step
                    self._m_max = max(self._m_max, step)
        elif self.on_target == "restart":
            self._m_goals += sum(was_on_goal)
        if not finished:
            return None
        if self.on_target == "finish":
            unsolved = n - self._m_solved
            total = self._m_sum + unsolved * step
            mx = step if unsolved else self._m_max
            return {"ISR": self._m_solved / n, "CSR": float(self._m_solved == n), "ep_length": total / n + 1,
                    "SoC": float(total + n), "makespan": float(mx + 1), "avg_throughput": 0.0}
        if self.on_target == "nothing":
            on = sum(was_on_goal)
            return {"ISR": on / n, "CSR": float(on == n), "ep_length": float(step + 1), "SoC": float(n * (step + 1)),
                    "makespan": float(step + 1), "avg_throughput": 0.0}
        denom = self.max_episode_steps if self.max_episode_steps > 0 else step + 1
        return {"ISR": 0.0, "CSR": 0.0, "ep_length": float(step + 1), "SoC": 0.0, "makespan": 0.0,
                "avg_throughput": self._m_goals / denom}

    # -- POMAPF / MAPF dict observations (upstream `PogemaBase._pomapf_obs` / `_mapf_obs`) ----------
    def pomapf_obs(self, global_info=False):
        g = self.grid
        r = self.obs_radius
        starts = [(x + r, y + r) for x, y in self._init_args[1]]
        out = []
        for i in range(self.num_agents):
            x0, y0 = starts[i]
            d = {"obstacles": g.get_obstacles_for_agent(i), "agents": g.get_positions(i),
                 "xy": (g.positions_xy[i][0] - x0, g.positions_xy[i][1] - y0),
                 "target_xy": (g.finishes_xy[i][0] - x0, g.finishes_xy[i][1] - y0)}
            if global_info:
                d["global_obstacles"] = g.obstacles[r:-r, r:-r].astype(np.float32)
                d["global_xy"] = (g.positions_xy[i][0] - r, g.positions_xy[i][1] - r)
                d["global_target_xy"] = (g.finishes_xy[i][0] - r, g.finishes_xy[i][1] - r)
            out.append(d)
        return out

    def _generate_new_target(self, agent_idx):
        r = self.obs_radius
        x, y = self.grid.positions_xy[agent_idx]
        comp = self._comp_points[self._labels[x - r, y - r]]
        if self.lifelong_rng == "numpy":
            # upstream generate_new_target: `tuple(*rnd_generator.choice(component, 1))` (the component's cell ORDER is
            # build-defined: row-major)
            tx, ty = (int(v) for v in self._random_generators[agent_idx].choice(comp, 1)[0])
            self._target_counter[agent_idx] += 1
            return (tx + r, ty + r)
        k = lifelong_draw(self.seed, self.env_index, agent_idx, self._target_counter[agent_idx], len(comp))
        self._target_counter[agent_idx] += 1
        tx, ty = comp[k]
        return (tx + r, ty + r)

    def set_targets(self, targets_xy, mask=None):
        """Overwrite the current targets (unpadded coordinates) of the agents flagged in `mask` (None = all): replay
        of a recorded lifelong target sequence (tests/test_golden_reference.py)."""
        r = self.obs_radius
        for i, (x, y) in enumerate(targets_xy):
            if mask is None or mask[i]:
                self.grid.finishes_xy[i] = (int(x) + r, int(y) + r)

    # -- state export (unpadded coordinates), used by parity tests ---------------------------------
    def get_state(self):
        g = self.grid
        return {
            "agents_xy": np.array(g.unpadded_xy(g.positions_xy), dtype=np.int32).reshape(-1, 2),
            "targets_xy": np.array(g.unpadded_xy(g.finishes_xy), dtype=np.int32).reshape(-1, 2),
            "is_active": np.array([g.is_active[i] for i in range(self.num_agents)], dtype=np.uint8),
            "elapsed": self._elapsed_steps,
            "occupancy": g.positions.astype(np.uint8).copy(),
        }
