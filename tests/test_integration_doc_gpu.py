"""GPU: the ctypes stub printed in INTEGRATION.md section 1 (what a maintainer of the reference would add) is real
code: it is extracted from the document, executed, and driven with stand-in `Grid` objects against the oracle."""
import os
import re

import numpy as np
import pytest

from oracle.pogema_oracle import PogemaOracle
from util import generate_instances

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeGrid:
    """Duck-types the three accessors of the reference's Grid the stub calls."""

    def __init__(self, obstacles, agents, targets):
        self._o, self._a, self._t = obstacles, agents, targets

    def get_obstacles(self, ignore_borders=False):
        return self._o

    def get_agents_xy(self, ignore_borders=False):
        return self._a

    def get_targets_xy(self, ignore_borders=False):
        return self._t


def test_documented_ctypes_stub_runs():
    import torch
    from pogema_amd import GridConfig
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libpogema_amd.so")', f'C.CDLL("{os.path.join(ROOT, "pogema_amd", "libpogema_amd.so")}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    B, S, A, r = 5, 10, 4, 2
    obstacles, agents, targets = generate_instances(B, S, S, A, 0.2, 3)
    gc = GridConfig(size=S, num_agents=A, obs_radius=r, density=0.2, seed=1, collision_system="soft")
    env = ns["AmdVecPogema"](gc, B)
    obs = env.reset_from_grids([_FakeGrid(obstacles[b], agents[b], targets[b]) for b in range(B)])
    refs = [PogemaOracle(obstacles[b], agents[b], targets[b], obs_radius=r, collision_system="soft") for b in range(B)]
    assert np.array_equal(obs.cpu().numpy(), np.stack([np.stack(e._obs()) for e in refs]))
    rng = np.random.default_rng(0)
    for _ in range(6):
        acts = rng.integers(0, 5, size=(B, A))
        o, rew, term, trunc, info = env.step(torch.from_numpy(acts).cuda())
        for b, e in enumerate(refs):
            ro, rr, rt, rtr, ri = e.step(acts[b])
            assert np.array_equal(o[b].cpu().numpy(), np.stack(ro)) and rew[b].tolist() == rr and term[b].tolist() == rt
