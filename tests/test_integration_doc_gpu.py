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


def _documented_stub():
    """The code block of INTEGRATION.md section 1, executed: returns its namespace."""
    import torch  # noqa: F401  (the block expects torch's HIP runtime to be the one in the process)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libpogema_amd.so")', f'C.CDLL("{os.path.join(ROOT, "pogema_amd", "libpogema_amd.so")}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    return ns


def test_documented_ctypes_stub_runs():
    import torch
    from pogema_amd import GridConfig
    ns = _documented_stub()
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
    env.close()


def _us_per_step(step, actions, steps=200, windows=3):
    import torch
    best = float("inf")
    for _ in range(windows + 1):  # the first window warms up
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for t in range(steps):
            step(actions[t % len(actions)])
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / steps * 1e3)
    return best


def test_documented_stub_lands_on_the_fast_tier():
    """a maintainer who copies INTEGRATION.md section 1 must get the engine's fast tier, not the one of
    `torch.empty` per step.  BASELINE configs[2] (8192 x 64x64 x 64 agents): 200 steps of the documented stub against
    VecPogema (explicit walk, same budget) on the same box, and against fresh torch tensors per step.  The 5 % bound is
    asserted when both found a second HBM zone (`spread`) AND the stub's one pool delivered what its walk promised; boxes
    without zones run every variant at the slow rate, and a pool whose fast stretch was narrower than its buffers is the
    case the stub documents and leaves to the shipped mirror (which retries further along the walk)."""
    import ctypes as C
    import torch
    from pogema_amd import GridConfig, VecPogema, _lib
    ns = _documented_stub()
    B, S, A, r = 8192, 64, 64, 5
    obstacles, agents, targets = generate_instances(B, S, S, A, 0.3, 0)
    gc = GridConfig(size=S, num_agents=A, obs_radius=r, density=0.3, seed=0, collision_system="soft", max_episode_steps=64)
    gen = torch.Generator(device="cuda").manual_seed(1)
    actions = [torch.randint(0, 5, (B, A), generator=gen, device="cuda") for _ in range(8)]  # int64, as the stub passes them

    stub = ns["AmdVecPogema"](gc, B, auto_reset=1)
    first = stub.reset_from_grids([_FakeGrid(obstacles[b], agents[b], targets[b]) for b in range(B)]).clone()  # (a rotating buffer)
    info = _lib.PgxBuffersInfo()
    ns["lib"].pgx_buffers_get_info(stub.pool, C.byref(info))
    stub_us = _us_per_step(stub.step, actions)

    env = VecPogema(gc, batch=B, auto_reset=True, placement_budget_gib="half")
    obs0 = env.reset_from_state(obstacles, agents, targets, validate=False)
    assert torch.equal(first, obs0)
    env.warm_buffers()
    ours_us = _us_per_step(env.step, actions)
    # same state afterwards: both ran the same 800 steps from the same instances
    o_stub = stub.step(actions[0])[0]
    o_ours = env.step(actions[0])[0]
    assert torch.equal(o_stub, o_ours), "the documented stub and VecPogema disagree after 801 steps"
    spread_ours = bool(env.placement.get("spread"))
    env.close(release=True)
    del o_ours, obs0, first

    plain = VecPogema(gc, batch=B, auto_reset=True, reuse_buffers=False)
    plain.reset_from_state(obstacles, agents, targets, validate=False)
    plain_us = _us_per_step(plain.step, actions)
    plain.close()
    print(f"\nconfigs[2], us per step: documented stub {stub_us:.1f} (spread={info.spread}), VecPogema {ours_us:.1f} "
          f"(spread={spread_ours}), fresh torch tensors per step {plain_us:.1f}")
    # the stub builds ONE pool; where the fast stretch the walk found is narrower than its buffers (the kept observation
    # passes miss the walk's promise, scaled from the 2 x 384 MiB probe to this tensor) it lands on the torch-placed tier --
    # VecPogema retries further along the walk in that case, the documented stub says so and does not
    promise_us = info.final_us * (B * A * 3 * 121 * 4) / (2 * (384 << 20))
    kept = max(stub.observe_us)
    print(f"stub's kept observation passes {[round(u, 1) for u in stub.observe_us]} us against the walk's promise {promise_us:.1f}")
    if info.spread and spread_ours and kept <= 1.10 * promise_us:
        assert stub_us <= 1.05 * ours_us, f"the documented binding runs {stub_us:.1f} us per step, VecPogema {ours_us:.1f}"
    stub.close()
