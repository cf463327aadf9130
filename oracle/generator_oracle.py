"""CPU ORACLE (test infrastructure only) -- literal Python restatement of the instance generator.

    *** PARITY UNPINNED w.r.t. upstream ***
    Upstream `pogema/generator.py` (`generate_obstacles`, `generate_positions_and_targets_fast`; SURVEY.md
    L1 / section 8f rank 2) draws from numpy's PCG64 `Generator` (`binomial`, `shuffle`); that stream cannot be
    reproduced on a GPU and the source is not mounted (/root/reference/README.md:3,5), so the build defines
    its OWN counter-based generator with the same contract: Bernoulli(density) obstacles; starts and targets
    on distinct free cells; every start/target pair inside one 4-connected component; OverflowError-style
    failure when the agents cannot be placed.  This file is the normative statement of that generator
    ("GEN v2"); oracle/pogema_oracle.c (po_generate), the host generator (pgx_generate) and the device
    kernels (pgx_reset_random) must all equal it bit for bit.

GEN v2, for the instance of global env index `env`, generation `epoch`, attempt `k`:
    h        = splitmix64(splitmix64(splitmix64(seed) ^ env) ^ (epoch << 32 | k))
    obstacle(c)  <=>  (splitmix64(h ^ (TAG_OBST | c)) >> 40) < thr,   thr = floor(density * 2^24 + 0.5)
                      for the row-major cell index c
    label(c)     =   smallest row-major index of c's 4-connected component of FREE cells
    placement: walk the candidate stream  c_t = ((splitmix64(h ^ (TAG_PLACE | t)) >> 32) * cells) >> 32,
               t = 0, 1, ...; skip obstacles and cells already taken; the first visit of a component
               opens a pair (its start), the next visit of the same component closes it (its target);
               agents are numbered in the order their pairs close; stop after `num_agents` pairs;
               give up (next attempt: k + 1, new obstacles unless the map is given) after 32 * cells + 64
               candidates.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

from .pogema_oracle import splitmix64

_MASK64 = 0xFFFFFFFFFFFFFFFF
TAG_OBST = 0x4F42535400000000   # 'OBST'
TAG_PLACE = 0x504C414300000000  # 'PLAC'


def density_threshold(density: float) -> int:
    return int(np.floor(float(np.float32(density)) * 16777216.0 + 0.5))


def instance_hash(seed: int, env: int, epoch: int, attempt: int) -> int:
    h = splitmix64(seed & _MASK64)
    h = splitmix64(h ^ (env & _MASK64))
    return splitmix64(h ^ (((epoch & 0xFFFFFFFF) << 32) | (attempt & 0xFFFFFFFF)))


def draw_obstacles(h: int, height: int, width: int, thr: int) -> np.ndarray:
    out = np.zeros((height, width), np.uint8)
    for c in range(height * width):
        if (splitmix64(h ^ (TAG_OBST | c)) >> 40) < thr:
            out[c // width, c % width] = 1
    return out


def min_index_labels(obstacles: np.ndarray) -> np.ndarray:
    """label(c) = smallest row-major index in c's 4-connected free component; -1 on obstacles."""
    height, width = obstacles.shape
    labels = -np.ones(height * width, np.int64)
    flat = obstacles.reshape(-1)
    for s in range(height * width):
        if flat[s] != 0 or labels[s] >= 0:
            continue
        labels[s] = s  # row-major scan: the first cell of a component reached is its smallest index
        stack = [s]
        while stack:
            c = stack.pop()
            x, y = divmod(c, width)
            for nx, ny in ((x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)):
                if 0 <= nx < height and 0 <= ny < width:
                    n = nx * width + ny
                    if flat[n] == 0 and labels[n] < 0:
                        labels[n] = s
                        stack.append(n)
    return labels.reshape(height, width)


def place_pairs(h: int, obstacles: np.ndarray, num_agents: int):
    """Returns (agents_xy, targets_xy) int32 [A, 2] or None when the candidate budget runs out."""
    height, width = obstacles.shape
    cells = height * width
    labels = min_index_labels(obstacles).reshape(-1)
    flat = obstacles.reshape(-1)
    taken = set()
    pending = {}
    agents, targets = [], []
    for t in range(32 * cells + 64):
        if len(agents) == num_agents:
            break
        c = ((splitmix64(h ^ (TAG_PLACE | t)) >> 32) * cells) >> 32
        if flat[c] != 0 or c in taken:
            continue
        taken.add(c)
        root = int(labels[c])
        if root not in pending:
            pending[root] = c
        else:
            s = pending.pop(root)
            agents.append(divmod(s, width))
            targets.append(divmod(c, width))
    if len(agents) < num_agents:
        return None
    return np.array(agents, np.int32), np.array(targets, np.int32)


def generate_instance(seed: int, env: int, height: int, width: int, num_agents: int, density: float,
                      epoch: int = 0, max_retries: int = 10, given_map=None):
    """One instance; raises OverflowError like the reference when the agents cannot be placed."""
    if 2 * num_agents > height * width:
        raise OverflowError("more start/target cells requested than the map has")
    thr = density_threshold(density)
    for attempt in range(max_retries):
        h = instance_hash(seed, env, epoch, attempt)
        obstacles = np.asarray(given_map, np.uint8) if given_map is not None else draw_obstacles(h, height, width, thr)
        placed = place_pairs(h, obstacles, num_agents)
        if placed is not None:
            return obstacles, placed[0], placed[1]
    raise OverflowError(f"could not place {num_agents} agents after {max_retries} attempts")


def generate_batch(seed, batch, height, width, num_agents, density, env_index_base=0, epoch=0, max_retries=10,
                   given_map=None):
    out = [generate_instance(seed, env_index_base + b, height, width, num_agents, density, epoch, max_retries, given_map)
           for b in range(batch)]
    return (np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out]))


TAG_POSSIBLE_AGENTS = 0x5041475400000000   # 'PAGT'
TAG_POSSIBLE_TARGETS = 0x5054475400000000  # 'PTGT'


def place_from_possible(seed, env, possible_agents_xy, possible_targets_xy, num_agents, epoch=0):
    """`GridConfig.possible_agents_xy / possible_targets_xy` (upstream `generate_from_possible_positions`: shuffle
    both lists, take the first num_agents of each -- numpy stream, not reproducible).  Build-defined: the first
    `num_agents` DISTINCT entries of the candidate stream idx_t = hash(h, tag, t) scaled to the list length."""
    if len(possible_agents_xy) < num_agents or len(possible_targets_xy) < num_agents:
        raise OverflowError("not enough possible positions")
    h = instance_hash(seed, env, epoch, 0)
    out = []
    for tag, cells in ((TAG_POSSIBLE_AGENTS, possible_agents_xy), (TAG_POSSIBLE_TARGETS, possible_targets_xy)):
        n, chosen, t = len(cells), [], 0
        while len(chosen) < num_agents:
            if t >= 32 * n + 64:
                raise OverflowError("candidate budget exhausted")
            i = ((splitmix64(h ^ (tag | t)) >> 32) * n) >> 32
            t += 1
            if i not in chosen:
                chosen.append(i)
        out.append(np.array([cells[i] for i in chosen], np.int32).reshape(-1, 2))
    return out[0], out[1]


def generate_instance_numpy(seed, height, width, num_agents, density, given_map=None):
    """Upstream's generator as RECALLED (pogema/generator.py `generate_obstacles`, `generate_positions_and_targets_fast`,
    `placing`; conf. medium -- the source is not mounted), written with numpy ITSELF: the checker of
    pgx_np_generate / pgx_np_generate_host.  Raises OverflowError when fewer than `num_agents` pairs exist."""
    from .pogema_oracle import label_components
    if given_map is not None:
        obstacles = (np.asarray(given_map) != 0).astype(np.int64)
    else:
        obstacles = np.random.default_rng(seed).binomial(1, density, (height, width))
    labels, _ = label_components(obstacles)
    order = [(x, y) for x in range(height) for y in range(width) if obstacles[x, y] == 0]
    np.random.default_rng(seed).shuffle(order)
    link_to_next = [-1] * len(order)
    colors = {}
    for index in range(len(order)):
        reversed_index = len(order) - index - 1
        color = int(labels[order[reversed_index]])
        link_to_next[reversed_index] = colors.get(color, -1)
        colors[color] = reversed_index
    positions_xy, finishes_xy = [], []
    for index in range(len(order)):
        next_index = link_to_next[index]
        if next_index == -1:
            continue
        positions_xy.append(order[index])
        finishes_xy.append(order[next_index])
        link_to_next[next_index] = -1
        if len(finishes_xy) >= num_agents:
            break
    if len(finishes_xy) < num_agents:
        raise OverflowError(f"only {len(finishes_xy)} of {num_agents} start/target pairs can be placed")
    return obstacles.astype(np.uint8), np.array(positions_xy, np.int32), np.array(finishes_xy, np.int32)


def policy_action(seed: int, env: int, agent: int, step: int) -> int:
    """The engine's uniform random policy (pgx_rollout with actions = NULL; pgx_kernels.hip: policy_action)."""
    h = splitmix64((seed ^ 0x504F4C4943590000) & _MASK64)
    h = splitmix64(h ^ (env & _MASK64))
    h = splitmix64(h ^ (((step << 20) | agent) & _MASK64))
    return ((h >> 32) * 5) >> 32


def policy_actions(seed: int, env_index_base: int, batch: int, agents: int, step0: int, steps: int) -> np.ndarray:
    return np.array([[[policy_action(seed, env_index_base + b, a, step0 + t) for a in range(agents)] for b in range(batch)]
                     for t in range(steps)], dtype=np.int8)
