"""GPU: the HIP engine against the hand-derived SPEC vectors (tests/golden/spec_vectors.json) --
expectations that were produced by neither the oracle nor the engine, including both values of every switch for the
low-confidence recollections (docs/SPEC.md Q1 / Q4 / Q7)."""
import json
import os

import pytest

from util import check_spec_case, engine_rollout

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "spec_vectors.json")) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_engine_matches_spec_vector(case):
    check_spec_case(case, engine_rollout)
