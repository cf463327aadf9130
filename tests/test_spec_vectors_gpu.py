"""GPU: the HIP engine against the hand-derived SPEC vectors (tests/golden/spec_vectors.json) --
expectations that were produced by neither the oracle nor the engine."""
import json
import os

import numpy as np
import pytest

from util import engine_rollout

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "spec_vectors.json")) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_engine_matches_spec_vector(case):
    obstacles = np.array(case["map"], np.uint8)[None]
    agents = np.array(case["agents_xy"], np.int32)[None]
    targets = np.array(case["targets_xy"], np.int32)[None]
    actions = np.array(case["actions"], np.int64)[:, None, :]
    out = engine_rollout(obstacles, agents, targets, actions, obs_radius=case["obs_radius"],
                         collision_system=case["collision_system"], on_target=case["on_target"],
                         max_episode_steps=case.get("max_episode_steps", 64), auto_reset=False)
    exp = case["expect"]
    assert out["agents_xy"][:, 0].tolist() == exp["agents_xy"], case["why"]
    if "rewards" in exp:
        assert out["rewards"][:, 0].tolist() == exp["rewards"]
    for key in ("terminated", "truncated", "is_active"):
        if key in exp:
            assert out[key][:, 0].astype(int).tolist() == exp[key], key
    if "obs0_agent0" in exp:
        assert out["obs0"][0, 0].astype(int).tolist() == exp["obs0_agent0"]
