"""Host-side (numpy) placement from `GridConfig.possible_agents_xy` / `possible_targets_xy`.

Replaces upstream `pogema/generator.py: generate_from_possible_positions` (recalled: shuffle both lists with the
grid's numpy generator, take the first `num_agents` of each; `OverflowError` when a list is too short).  The
numpy stream is not reproduced; the build walks its own counter-based candidate streams -- generator
GEN v2's instance hash with the tags 'PAGT' / 'PTGT' -- and takes the first `num_agents` DISTINCT entries of each
list.  Normative statement: oracle/generator_oracle.py `place_from_possible`.  This is reset-path host code used
only with that (rare) GridConfig option; everything else is generated on the device.
"""
from __future__ import annotations

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
TAG_POSSIBLE_AGENTS = 0x5041475400000000   # 'PAGT'
TAG_POSSIBLE_TARGETS = 0x5054475400000000  # 'PTGT'


def _sm64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _instance_hash(seed: int, env: int, epoch: int = 0, attempt: int = 0) -> np.uint64:
    h = _sm64(np.array([seed & 0xFFFFFFFFFFFFFFFF], np.uint64))
    h = _sm64(h ^ np.uint64(env & 0xFFFFFFFFFFFFFFFF))
    return _sm64(h ^ np.uint64(((epoch & 0xFFFFFFFF) << 32) | (attempt & 0xFFFFFFFF)))[0]


def _first_distinct(h: np.uint64, tag: int, n: int, count: int) -> np.ndarray:
    """Indices into a list of length n: the first `count` distinct values of the candidate stream."""
    budget = 32 * n + 64
    chosen, seen, t0 = [], set(), 0
    while len(chosen) < count and t0 < budget:
        t = np.arange(t0, min(budget, t0 + 4 * count + 64), dtype=np.uint64)
        r = _sm64(h ^ (np.uint64(tag) | t))
        idx = ((r >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)
        for i in idx.tolist():
            if i not in seen:
                seen.add(i)
                chosen.append(i)
                if len(chosen) == count:
                    break
        t0 += len(t)
    if len(chosen) < count:
        raise OverflowError(f"could not draw {count} distinct entries out of {n}")
    return np.asarray(chosen, np.int64)


def place_from_possible(batch: int, seed0: int, possible_agents_xy, possible_targets_xy, num_agents: int,
                        env_index_base: int = 0):
    """agents_xy, targets_xy int32 [batch, A, 2]; env b draws from instance (seed0, env_index_base + b)."""
    pa = np.asarray(possible_agents_xy, np.int32).reshape(-1, 2)
    pt = np.asarray(possible_targets_xy, np.int32).reshape(-1, 2)
    if len(pa) < num_agents or len(pt) < num_agents:
        raise OverflowError(f"{num_agents} agents need at least as many possible start and target cells "
                            f"(got {len(pa)} and {len(pt)})")
    agents = np.empty((batch, num_agents, 2), np.int32)
    targets = np.empty((batch, num_agents, 2), np.int32)
    for b in range(batch):
        h = _instance_hash(seed0, env_index_base + b)
        agents[b] = pa[_first_distinct(h, TAG_POSSIBLE_AGENTS, len(pa), num_agents)]
        targets[b] = pt[_first_distinct(h, TAG_POSSIBLE_TARGETS, len(pt), num_agents)]
    return agents, targets
