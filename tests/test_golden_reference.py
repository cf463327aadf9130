"""Checks oracle (CPU) and engine (GPU) against REAL reference fixtures tests/golden/reference_*.npz
when they exist (made by tools/gen_golden.py where `pogema` is importable).  None can exist in this
build container (the reference is unavailable) -> parity stays 'unpinned' and these tests skip."""
import glob
import os

import numpy as np
import pytest

from util import assert_rollouts_equal, engine_rollout, oracle_rollout

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "reference_*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    ref = {k: z[k] for k in ("obs0", "obs", "rewards", "terminated", "truncated", "is_active", "agents_xy", "targets_xy")}
    T = z["actions"].shape[0]
    for k in ("obs", "rewards", "terminated", "truncated", "is_active", "agents_xy", "targets_xy"):
        ref[k] = ref[k][:, None]
    ref["obs0"] = ref["obs0"][None]
    ref["elapsed"] = np.arange(1, T + 1, dtype=np.int32)[:, None]
    kw = dict(obs_radius=int(z["obs_radius"]), collision_system=str(z["collision_system"]), on_target=str(z["on_target"]),
              max_episode_steps=int(z["max_episode_steps"]), auto_reset=False)
    return z["obstacles"][None], z["agents_xy0"][None], z["targets_xy0"][None], z["actions"][:, None, :], ref, kw


@pytest.mark.skipif(not FIXTURES, reason="no reference fixtures: the reference is not available in this container")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_matches_reference_fixture(path):
    obstacles, agents, targets, actions, ref, kw = _load(path)
    if kw["on_target"] == "restart":
        ref.pop("targets_xy")  # lifelong target stream is build-defined (DESIGN.md open question 5)
    got = oracle_rollout(obstacles, agents, targets, actions, **kw)
    for k in list(got):
        if k not in ref:
            got.pop(k)
    assert_rollouts_equal({**got, **ref}, got, os.path.basename(path))


@pytest.mark.gpu
@pytest.mark.skipif(not FIXTURES, reason="no reference fixtures: the reference is not available in this container")
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_engine_matches_reference_fixture(path):
    obstacles, agents, targets, actions, ref, kw = _load(path)
    if kw["on_target"] == "restart":
        pytest.skip("lifelong target stream is build-defined")
    got = engine_rollout(obstacles, agents, targets, actions, **kw)
    assert_rollouts_equal(ref, got, os.path.basename(path))
