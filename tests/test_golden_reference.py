"""Checks oracle (CPU) and engine (GPU) against REAL reference fixtures tests/golden/reference_*.npz when they exist
(made by tools/gen_golden.py where `pogema` is importable).  None can exist in this build container (the reference is
unavailable) -> parity stays 'unpinned' and these tests skip; tests/test_golden_pipeline.py runs the very same loader
and comparison against fixtures generated from a stand-in package, so the procedure is known to work.

Lifelong (`on_target='restart'`) fixtures: the reference draws every new target from per-agent numpy generators, a
stream the implementations do not reproduce (docs/SPEC.md S5).  The recorded target sequence is therefore REPLAYED:
after every step the implementation's targets are overwritten with the fixture's (`pgx_set_targets`), so positions,
rewards, flags and observations stay comparable for the whole episode.  Only the draw itself is excluded: for a
(step, agent) that reached its goal in that step the new target and the target plane of that step's observation are
taken from the fixture.

Instance generator: a fixture that records `grid_seed` and `density` also pins the numpy-stream generator
(pgx_np_generate / Semantics.generator_rng='numpy'): the instance it draws for that seed must be the fixture's initial
state.  A failure of THAT test alone means the recalled `placing` rule or call order differs from the real package --
the step semantics are unaffected (every other test starts from the recorded initial state)."""
import glob
import os

import numpy as np
import pytest

from util import assert_rollouts_equal, engine_rollout, oracle_rollout

# PGX_GOLDEN_DIR: fixtures somewhere else than tests/golden (tools/pin_reference.sh --out DIR; the stand-in rehearsal)
GOLDEN_DIR = os.environ.get("PGX_GOLDEN_DIR") or os.path.join(os.path.dirname(__file__), "golden")
GRID_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "reference_grid_*.npz")))  # tools/gen_golden_grid.py: the grid layer alone
FIXTURES = [p for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "reference_*.npz"))) if p not in GRID_FIXTURES]
KEYS = ("obs0", "obs", "rewards", "terminated", "truncated", "is_active", "agents_xy", "targets_xy")
METRIC_NAMES = ("ISR", "CSR", "ep_length", "SoC", "makespan", "avg_throughput")  # column order of the engine's metrics


def _load(path):
    z = np.load(path, allow_pickle=False)
    ref = {k: z[k] for k in KEYS}
    T = z["actions"].shape[0]
    for k in KEYS[1:]:
        ref[k] = ref[k][:, None]  # batch axis: one environment per fixture
    ref["obs0"] = ref["obs0"][None]
    ref["elapsed"] = np.arange(1, T + 1, dtype=np.int32)[:, None]
    if "positions" in z.files:  # round 4 fixtures: the occupancy array itself (`grid.positions`, padded)
        ref["occupancy0"] = z["positions0"][None].astype(np.uint8)
        ref["occupancy"] = z["positions"][:, None].astype(np.uint8)
    if "metrics_names" in z.files:  # infos[0]['metrics'] of the step that ended the episode: only the names both sides know
        row = dict(zip(str(z["metrics_names"]).split("|"), np.asarray(z["metrics_values"], dtype=np.float64)))
        ref["metrics_final"] = (int(z["metrics_step"]), row)
    kw = dict(obs_radius=int(z["obs_radius"]), collision_system=str(z["collision_system"]), on_target=str(z["on_target"]),
              max_episode_steps=int(z["max_episode_steps"]), auto_reset=False)
    return z["obstacles"][None], z["agents_xy0"][None], z["targets_xy0"][None], z["actions"][:, None, :], ref, kw


def compare_with_fixture(run, path):
    """`run` = oracle_rollout or engine_rollout; raises AssertionError on the first difference."""
    obstacles, agents, targets, actions, ref, kw = _load(path)
    lifelong = kw["on_target"] == "restart"
    extra = {"with_occupancy": True} if ("occupancy" in ref and run is engine_rollout) else {}
    got = run(obstacles, agents, targets, actions, inject_targets=ref["targets_xy"] if lifelong else None, **kw, **extra)
    final = ref.pop("metrics_final", None)
    if final is not None:  # docs/SPEC.md Q9: the metric wrappers' formulas, for every name the reference reports
        step, row = final
        assert got["episode_done"][step, 0], f"{os.path.basename(path)}: the reference ended its episode in step {step}"
        for name, want in row.items():
            if name in METRIC_NAMES:
                have = float(got["metrics"][step, 0, METRIC_NAMES.index(name)])
                assert abs(have - want) <= 1e-6 * max(1.0, abs(want)), f"{os.path.basename(path)}: metric {name} = {have}, reference {want}"
    got = {k: v for k, v in got.items() if k in ref}
    if lifelong:
        drew = ref["rewards"] > 0  # [T, 1, A]: reached its goal in this step -> the reference drew a new target
        got["targets_xy"] = np.where(drew[..., None], ref["targets_xy"], got["targets_xy"])
        got["obs"] = got["obs"].copy()
        got["obs"][:, :, :, 2] = np.where(drew[..., None, None], ref["obs"][:, :, :, 2], got["obs"][:, :, :, 2])
    assert_rollouts_equal(ref, got, os.path.basename(path))


def compare_generator_with_fixture(generate, path):
    """`generate(seeds, H, W, A, density)` -> (obstacles, agents, targets, status) as numpy; False when the fixture
    does not record where its instance came from."""
    z = np.load(path, allow_pickle=False)
    if "grid_seed" not in z.files:
        return False
    H, W = z["obstacles"].shape
    o, a, t, st = (np.asarray(v) for v in generate([int(z["grid_seed"])], H, W, z["agents_xy0"].shape[0], float(z["density"])))
    what = os.path.basename(path)
    assert st[0] == 0, f"{what}: generator could not place the agents"
    np.testing.assert_array_equal(o[0], z["obstacles"], err_msg=f"{what}: obstacles")
    np.testing.assert_array_equal(a[0], z["agents_xy0"], err_msg=f"{what}: starts")
    np.testing.assert_array_equal(t[0], z["targets_xy0"], err_msg=f"{what}: targets")
    return True


@pytest.mark.skipif(not FIXTURES, reason="no reference fixtures: the reference is not available in this container")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_numpy_generator_matches_reference_fixture(path):
    from pogema_amd.nprng import np_generate_host
    if not compare_generator_with_fixture(np_generate_host, path):
        pytest.skip("fixture predates grid_seed/density")


@pytest.mark.skipif(not FIXTURES, reason="no reference fixtures: the reference is not available in this container")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_matches_reference_fixture(path):
    compare_with_fixture(oracle_rollout, path)


@pytest.mark.gpu
@pytest.mark.skipif(not FIXTURES, reason="no reference fixtures: the reference is not available in this container")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_engine_matches_reference_fixture(path):
    compare_with_fixture(engine_rollout, path)


# ---- fixtures of the grid layer alone (tools/gen_golden_grid.py: upstream's Grid driven without pogema/__init__.py) --------
def compare_grid_fixture(run, path):
    """A scripted episode of `grid.move(agent, action)` in index order IS collision_system='priority'; with
    on_target='nothing' nobody is ever hidden and no target changes, so positions, the occupancy array and all three
    observation planes of every agent after every step must match."""
    z = np.load(path, allow_pickle=False)
    T = z["actions"].shape[0]
    ref = {"obs0": z["obs0"][None], "obs": z["obs"][:, None], "agents_xy": z["agents_xy"][:, None]}
    extra = {}
    if "positions" in z.files:
        ref["occupancy0"] = z["positions0"][None].astype(np.uint8)
        ref["occupancy"] = z["positions"][:, None].astype(np.uint8)
        if run is engine_rollout:
            extra["with_occupancy"] = True
    got = run(z["obstacles"][None], z["agents_xy0"][None], z["targets_xy0"][None], z["actions"][:, None, :],
              obs_radius=int(z["obs_radius"]), collision_system="priority", on_target="nothing", max_episode_steps=T + 8,
              auto_reset=False, **extra)
    assert_rollouts_equal(ref, {k: v for k, v in got.items() if k in ref}, os.path.basename(path))


def compare_border_with_grid_fixture(path):
    """the padded obstacle array upstream's `Grid.__init__` builds (border ring at offset r - 1, free outside)
    against the oracle's."""
    from oracle.pogema_oracle import Grid
    z = np.load(path, allow_pickle=False)
    g = Grid(z["obstacles"], z["agents_xy0"], z["targets_xy0"], int(z["obs_radius"]))
    np.testing.assert_array_equal((g.obstacles != 0).astype(np.uint8), z["padded_obstacles"], err_msg=os.path.basename(path))


_no_grid = pytest.mark.skipif(not GRID_FIXTURES, reason="no grid-layer fixtures: the reference's source is not available in this container")


@_no_grid
@pytest.mark.parametrize("path", GRID_FIXTURES, ids=[os.path.basename(p) for p in GRID_FIXTURES])
def test_oracle_matches_grid_fixture(path):
    compare_grid_fixture(oracle_rollout, path)
    compare_border_with_grid_fixture(path)


@_no_grid
@pytest.mark.parametrize("path", GRID_FIXTURES, ids=[os.path.basename(p) for p in GRID_FIXTURES])
def test_numpy_generator_matches_grid_fixture(path):
    from pogema_amd.nprng import np_generate_host
    assert compare_generator_with_fixture(np_generate_host, path)


@pytest.mark.gpu
@_no_grid
@pytest.mark.parametrize("path", GRID_FIXTURES, ids=[os.path.basename(p) for p in GRID_FIXTURES])
def test_engine_matches_grid_fixture(path):
    compare_grid_fixture(engine_rollout, path)
