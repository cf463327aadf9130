"""CPU tests of the oracle itself (no GPU): the plain-C port must equal the literal Python
restatement bit for bit, and both must reproduce the hand-derived SPEC vectors in tests/golden/."""
import json
import os
import zlib

import numpy as np
import pytest

from util import (assert_rollouts_equal, c_oracle_rollout, check_spec_case, generate_instances, oracle_rollout,
                  random_actions)

COLLISIONS = ("priority", "block_both", "soft")
ON_TARGET = ("finish", "restart", "nothing")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

GEOMS = [
    ("cfg0", 4, 8, 8, 2, 3, 0.3, 20, 8),
    ("cfg1", 3, 16, 16, 8, 5, 0.3, 20, 8),
    ("dense", 4, 6, 6, 14, 2, 0.1, 30, 10),
    ("crowd", 2, 10, 10, 40, 4, 0.1, 25, 9),
    ("rect", 3, 5, 11, 9, 1, 0.15, 20, 7),
]


@pytest.mark.parametrize("geom", GEOMS, ids=[g[0] for g in GEOMS])
@pytest.mark.parametrize("collision", COLLISIONS)
@pytest.mark.parametrize("on_target", ON_TARGET)
def test_c_port_equals_python_oracle(geom, collision, on_target):
    name, B, H, Wd, A, r, density, T, max_steps = geom
    seed = zlib.crc32(f"{name}/{collision}/{on_target}".encode()) % (2 ** 31)
    obstacles, agents, targets = generate_instances(B, H, Wd, A, density, seed)
    actions = random_actions(T, B, A, seed + 1)
    for auto_reset in (False, True):
        kw = dict(obs_radius=r, collision_system=collision, on_target=on_target, max_episode_steps=max_steps,
                  auto_reset=auto_reset, seed=77, env_index_base=3)
        ref = oracle_rollout(obstacles, agents, targets, actions, **kw)
        got = c_oracle_rollout(obstacles, agents, targets, actions, **kw)
        assert_rollouts_equal(ref, got, f"{name}/{collision}/{on_target}/{auto_reset}")


@pytest.mark.parametrize("geom", [g for g in GEOMS if g[0] in ("dense", "crowd", "rect")], ids=lambda g: g[0])
@pytest.mark.parametrize("on_target", ON_TARGET)
def test_c_port_equals_python_oracle_semantics_variants(geom, on_target):
    """docs/SPEC.md Q1 / Q4 alternatives (soft 'all_stay', coop 'per_agent'): the two oracles agree there too, and the
    variants really differ from the defaults on crowded instances."""
    from pogema_amd.semantics import Semantics
    name, B, H, Wd, A, r, density, T, max_steps = geom
    seed = zlib.crc32(f"variants/{name}/{on_target}".encode()) % (2 ** 31)
    obstacles, agents, targets = generate_instances(B, H, Wd, A, density, seed)
    actions = random_actions(T, B, A, seed + 1)
    sem = Semantics(soft_vertex="all_stay", coop_reward="per_agent")
    kw = dict(obs_radius=r, collision_system="soft", on_target=on_target, max_episode_steps=max_steps, auto_reset=True,
              seed=77, env_index_base=3)
    ref = oracle_rollout(obstacles, agents, targets, actions, semantics=sem, **kw)
    got = c_oracle_rollout(obstacles, agents, targets, actions, semantics=sem, **kw)
    assert_rollouts_equal(ref, got, f"variants/{name}/{on_target}")
    base = oracle_rollout(obstacles, agents, targets, actions, **kw)
    if name in ("dense", "crowd"):
        assert not np.array_equal(base["agents_xy"], ref["agents_xy"]), "all_stay must differ from lowest_index somewhere"


def test_c_port_threads_deterministic():
    B, H, Wd, A, r = 16, 12, 12, 20, 3
    obstacles, agents, targets = generate_instances(B, H, Wd, A, 0.2, 4)
    actions = random_actions(10, B, A, 9)
    kw = dict(obs_radius=r, collision_system="soft", on_target="restart", max_episode_steps=6, auto_reset=True)
    a = c_oracle_rollout(obstacles, agents, targets, actions, nthreads=1, **kw)
    b = c_oracle_rollout(obstacles, agents, targets, actions, nthreads=4, **kw)
    assert_rollouts_equal(a, b, "threads")


def _load_spec_vectors():
    with open(os.path.join(GOLDEN, "spec_vectors.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _load_spec_vectors(), ids=lambda c: c["name"])
@pytest.mark.parametrize("impl", ["python", "c"])
def test_spec_vectors(case, impl):
    """Hand-derived expectations (SURVEY section 8a SPEC column) -- NOT outputs of the reference."""
    check_spec_case(case, oracle_rollout if impl == "python" else c_oracle_rollout)


def test_observation_planes_by_hand():
    """A9/A10/A11 on a 3x3 map with r=2: border ring, occupancy incl. self, clamped target."""
    from oracle.pogema_oracle import PogemaOracle
    obstacles = np.zeros((3, 3), np.uint8)
    obstacles[0, 1] = 1
    env = PogemaOracle(obstacles, [(1, 1), (2, 2)], [(1, 1 + 1), (0, 0)], obs_radius=2)
    obs = env._obs()
    # agent 0 at the map centre: window covers the whole padded 7x7 minus one ring -> 5x5
    exp_obst = np.array([[1, 1, 1, 1, 1], [1, 0, 1, 0, 1], [1, 0, 0, 0, 1], [1, 0, 0, 0, 1], [1, 1, 1, 1, 1]], np.float32)
    assert np.array_equal(obs[0][0], exp_obst)
    exp_pos = np.zeros((5, 5), np.float32)
    exp_pos[2, 2] = 1  # self
    exp_pos[3, 3] = 1  # agent 1
    assert np.array_equal(obs[0][1], exp_pos)
    exp_tgt = np.zeros((5, 5), np.float32)
    exp_tgt[2, 3] = 1  # target one cell to the right
    assert np.array_equal(obs[0][2], exp_tgt)
    # agent 1 at (2,2), target (0,0): offset (2,2) -> window cell (r-2, r-2) = (0, 0)
    assert obs[1][2][0, 0] == 1 and obs[1][2].sum() == 1
    # clamping: r=1 window for the same geometry puts the far target on the window corner
    env1 = PogemaOracle(obstacles, [(2, 2)], [(0, 0)], obs_radius=1)
    t = env1._obs()[0][2]
    assert t[0, 0] == 1 and t.sum() == 1


@pytest.mark.parametrize("r", [1, 3, 6])
def test_random_outside_c_port_equals_python(r):
    """`empty_outside=False`: Bernoulli obstacles beyond the border ring (build-defined stream, docs/SPEC.md S1)."""
    B, H, Wd, A = 3, 7, 9, 5
    obstacles, agents, targets = generate_instances(B, H, Wd, A, 0.2, 31)
    actions = random_actions(8, B, A, 2)
    kw = dict(obs_radius=r, collision_system="soft", on_target="finish", max_episode_steps=5, auto_reset=True, seed=77,
              env_index_base=3, empty_outside=False)
    ref = oracle_rollout(obstacles, agents, targets, actions, **kw)
    got = c_oracle_rollout(obstacles, agents, targets, actions, **kw)
    assert_rollouts_equal(ref, got, f"random outside r={r}")
    plain = oracle_rollout(obstacles, agents, targets, actions, **{**kw, "empty_outside": True})
    if r > 1:
        assert not np.array_equal(plain["obs0"][:, :, 0], ref["obs0"][:, :, 0]), "the outside must be visible somewhere"
    # the ring and the map interior are never touched, so the dynamics are identical
    assert np.array_equal(plain["agents_xy"], ref["agents_xy"])


def test_odd_configurations_python_oracle_equals_c_port():
    """Differential fuzz over the corners random instance generators never visit (tests/util.odd_cases): maps 1..6 cells wide
    (also a single row or column), agents that START on their goal, several agents sharing one goal, goals on other agents'
    start cells, windows larger than the map, time limits of 1-3 steps, out-of-range actions, every semantics switch,
    `empty_outside=False`, lifelong streams keyed by odd (seed, env index) pairs -- the literal Python oracle against the
    plain-C port; tests/test_fuzz_gpu.py runs the same cases through the engine.  (A time limit <= 0 is NOT in here: the C-ABI
    defines it as 'no limit', the Python surface maps it to the literal behaviour -- test_time_limit_edge_cases.)"""
    from util import assert_rollouts_equal, c_oracle_rollout, odd_cases, oracle_rollout
    for what, args, kw in odd_cases(20261002, 220):
        assert_rollouts_equal(oracle_rollout(*args, **kw), c_oracle_rollout(*args, **kw), what)
