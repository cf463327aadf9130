"""PipelinedVecPogema (S engines on S streams over one batch) yields exactly what ONE engine over the whole batch
yields: same instances, same per-step outputs, same lifelong targets -- the split only changes when things run."""
import numpy as np
import pytest
import torch

from pogema_amd import GridConfig, PipelinedVecPogema, VecPogema
from util import random_actions

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("parts", [1, 2, 4])
@pytest.mark.parametrize("on_target", ["finish", "restart"])
def test_pipelined_equals_single_engine(parts, on_target):
    B, A, T = 24, 8, 30
    gc = GridConfig(size=16, num_agents=A, obs_radius=4, density=0.25, collision_system="soft", on_target=on_target,
                    max_episode_steps=12, seed=9)
    one = VecPogema(gc, batch=B, device=DEV, auto_reset=True, env_index_base=5)
    obs1, _ = one.reset(seed=3)
    pipe = PipelinedVecPogema(gc, batch=B, device=DEV, parts=parts, auto_reset=True, env_index_base=5)
    first = pipe.reset(seed=3)
    pipe.synchronize()
    assert torch.equal(torch.cat([o for o, _ in first]), obs1)
    actions = torch.as_tensor(random_actions(T, B, A, seed=2), device=DEV)
    for t in range(T):
        o1, r1, te1, tr1, i1 = one.step(actions[t])
        res = pipe.step(actions[t])
        pipe.synchronize()
        assert torch.equal(torch.cat([r[0] for r in res]), o1), f"obs at step {t}"
        assert torch.equal(torch.cat([r[1] for r in res]), r1)
        assert torch.equal(torch.cat([r[2] for r in res]), te1)
        assert torch.equal(torch.cat([r[3] for r in res]), tr1)
        assert torch.equal(torch.cat([r[4]["is_active"] for r in res]), i1["is_active"])
        assert torch.equal(torch.cat([r[4]["episode_done"] for r in res]), i1["episode_done"])
    s1, sp = one.get_state(), pipe.get_state()
    for k in s1:
        assert torch.equal(s1[k], sp[k]), k
    pipe.close()


def test_double_buffered_loop_with_a_policy_on_the_part_streams():
    """The intended use: each half's 'policy' runs on that half's stream, no host synchronisation inside the loop."""
    B, A = 64, 4
    gc = GridConfig(size=8, num_agents=A, obs_radius=2, density=0.1, max_episode_steps=16, seed=1)
    pipe = PipelinedVecPogema(gc, batch=B, device=DEV, parts=2, auto_reset=True, reuse_buffers=True)
    one = VecPogema(gc, batch=B, device=DEV, auto_reset=True)
    obs = [o for o, _ in pipe.reset(seed=0)]
    ref_obs, _ = one.reset(seed=0)

    def policy(o):  # deterministic function of the observation
        return (o.sum(dim=(2, 3, 4)).to(torch.int64) % 5).to(torch.int8)

    total = [torch.zeros((), device=DEV) for _ in range(2)]
    ref_total = torch.zeros((), device=DEV)
    for _ in range(40):
        for i in range(pipe.parts):
            with pipe.stream(i):
                obs[i], rew, *_ = pipe.step_part(i, policy(obs[i]))
                total[i] += rew.sum()
        ref_obs, rew, *_ = one.step(policy(ref_obs))
        ref_total += rew.sum()
    pipe.synchronize()
    assert float(total[0] + total[1]) == float(ref_total)
    assert torch.equal(torch.cat(obs), ref_obs)


def test_argument_errors():
    with pytest.raises(ValueError):
        PipelinedVecPogema(GridConfig(num_agents=2), batch=5, parts=2, device=DEV)
    pipe = PipelinedVecPogema(GridConfig(num_agents=2, seed=0), batch=4, parts=2, device=DEV)
    pipe.reset(seed=0)
    with pytest.raises(ValueError):
        pipe.step(torch.zeros((3, 2), dtype=torch.int8, device=DEV))
    with pytest.raises(ValueError):
        pipe.step([torch.zeros((2, 2), dtype=torch.int8, device=DEV)])


def test_obs_parents_rows_of_full_batch_tensors():
    """obs_parents: the parts write their rows of the caller's full-batch tensors in turn; results equal one engine's."""
    gc = GridConfig(size=16, num_agents=8, obs_radius=4, density=0.2, seed=5, collision_system="soft", max_episode_steps=12)
    B = 12
    one = VecPogema(gc, batch=B, device=DEV, auto_reset=True)
    one.reset(seed=5)
    parents = [torch.zeros(one.obs_shape, dtype=torch.float32, device=DEV) for _ in range(2)]
    pipe = PipelinedVecPogema(gc, batch=B, parts=2, device=DEV, auto_reset=True, obs_parents=parents)
    pipe.reset(seed=5)
    pipe.warm_buffers()
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    for t in range(15):
        acts = torch.randint(0, 5, (B, 8), generator=gen, device=DEV, dtype=torch.int8)
        ref = one.step(acts)
        res = pipe.step(acts)
        pipe.synchronize()
        obs, rew, term, trunc, act = pipe.parent_outputs(t % 2)
        assert obs.data_ptr() == parents[t % 2].data_ptr()
        assert torch.equal(obs, ref[0]) and torch.equal(rew, ref[1]) and torch.equal(term, ref[2]) and torch.equal(trunc, ref[3])
        assert torch.equal(act, ref[4]["is_active"])
        assert res[1][0].data_ptr() == obs[pipe.part_slice(1)].data_ptr()
    with pytest.raises(ValueError):
        PipelinedVecPogema(gc, batch=B, parts=2, device=DEV, obs_parents=[parents[0][:6]])
    pipe.close()
    one.close()
