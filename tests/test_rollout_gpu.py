"""pgx_rollout (K steps in one launch) must be indistinguishable from K pgx_step calls: every per-step output and the
final state, bit for bit -- across kernel variants (group sizes, multi-wave, helper waves, P16 / generic rows), all
collision systems and on_target modes, auto-reset, lifelong (both target streams), action dtypes and observation dtypes."""
import numpy as np
import pytest
import torch

from pogema_amd import GridConfig, Semantics, VecPogema
from util import generate_instances, oracle_rollout, random_actions, assert_rollouts_equal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(gc, batch, steps, seed=0, obs_slots=None, action_dtype=torch.int8, semantics=None, obs_dtype=None, **kw):
    extra = {} if obs_dtype is None else {"obs_dtype": obs_dtype}
    if semantics is not None:
        extra["semantics"] = semantics
    a = VecPogema(gc, batch=batch, device=DEV, **extra, **kw)
    b = VecPogema(gc, batch=batch, device=DEV, **extra, **kw)
    a.reset(seed=seed)
    b.reset(seed=seed)
    A = gc.num_agents
    actions = torch.as_tensor(random_actions(steps, batch, A, seed=seed + 1), device=DEV).to(action_dtype)
    ref = {k: [] for k in ("obs", "rewards", "terminated", "truncated", "is_active", "episode_done", "metrics")}
    for t in range(steps):
        obs, rew, term, trunc, infos = a.step(actions[t])
        ref["obs"].append(obs.clone())
        ref["rewards"].append(rew.clone())
        ref["terminated"].append(term.clone())
        ref["truncated"].append(trunc.clone())
        ref["is_active"].append(infos["is_active"].clone())
        ref["episode_done"].append(infos["episode_done"].clone())
        ref["metrics"].append(torch.where(infos["episode_done"][:, None], infos["metrics"], torch.zeros_like(infos["metrics"])))
    ref = {k: torch.stack(v) for k, v in ref.items()}
    got = b.rollout(actions, obs_slots=obs_slots)
    return a, b, ref, got


def _same(ref, got, steps, obs_slots=None):
    for k in ("rewards", "terminated", "truncated", "is_active", "episode_done"):
        assert torch.equal(ref[k], got[k]), k
    done = got["episode_done"]
    assert torch.equal(ref["metrics"], torch.where(done[..., None], got["metrics"], torch.zeros_like(got["metrics"])))
    if obs_slots is None:
        assert torch.equal(ref["obs"], got["obs"])
    elif obs_slots > 0:
        for t in range(max(0, steps - obs_slots), steps):
            assert torch.equal(ref["obs"][t], got["obs"][t % obs_slots]), f"obs slot of step {t}"


def _same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


GEOMS = [  # (size, agents, radius, batch): G = 2, 8, 16, 64, multi-wave 128 agents, helper waves (small batch of big envs), generic rows (r = 9)
    (8, 2, 3, 33), (16, 8, 5, 40), (32, 16, 5, 24), (64, 64, 5, 12), (40, 128, 4, 5), (64, 64, 5, 3), (24, 6, 9, 10),
    (256, 256, 7, 3)]  # ... and the geometry of BASELINE configs[4]


@pytest.mark.parametrize("size,agents,r,batch", GEOMS)
@pytest.mark.parametrize("collision", ["priority", "block_both", "soft"])
def test_rollout_equals_steps(size, agents, r, batch, collision):
    gc = GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.2, collision_system=collision, max_episode_steps=9,
                    seed=3)
    a, b, ref, got = _pair(gc, batch, 21, auto_reset=True)
    _same(ref, got, 21)
    _same_state(a, b)


@pytest.mark.parametrize("on_target", ["finish", "restart", "nothing"])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_rollout_modes(on_target, auto_reset):
    gc = GridConfig(size=12, num_agents=10, obs_radius=3, density=0.15, on_target=on_target, max_episode_steps=12, seed=5)
    a, b, ref, got = _pair(gc, 37, 30, auto_reset=auto_reset)
    _same(ref, got, 30)
    _same_state(a, b)


def test_rollout_numpy_lifelong_and_switches():
    gc = GridConfig(size=10, num_agents=6, obs_radius=2, density=0.1, on_target="restart", max_episode_steps=16, seed=2,
                    collision_system="soft")
    sem = Semantics(lifelong_rng="numpy", soft_vertex="all_stay")
    a, b, ref, got = _pair(gc, 19, 40, semantics=sem, auto_reset=True)
    _same(ref, got, 40)
    _same_state(a, b)


@pytest.mark.parametrize("slots", [0, 1, 2, 5])
def test_rollout_observation_ring(slots):
    gc = GridConfig(size=16, num_agents=8, obs_radius=4, density=0.2, max_episode_steps=8, seed=1)
    a, b, ref, got = _pair(gc, 16, 11, obs_slots=slots, auto_reset=True)
    _same(ref, got, 11, obs_slots=slots)
    if slots == 0:
        assert got["obs"] is None
    _same_state(a, b)
    # the handle keeps working step by step afterwards, and its observation is the state's
    assert torch.equal(a.observe(), b.observe())


@pytest.mark.parametrize("dt", [torch.int8, torch.int32, torch.int64])
def test_rollout_action_dtypes_and_uint8_obs(dt):
    gc = GridConfig(size=16, num_agents=8, obs_radius=5, density=0.2, max_episode_steps=8, seed=1)
    a, b, ref, got = _pair(gc, 9, 10, action_dtype=dt, obs_dtype=torch.uint8, auto_reset=True)
    _same(ref, got, 10)


def test_rollout_against_the_oracle():
    B, A, size, T = 6, 8, 16, 24
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, seed=11)
    actions = random_actions(T, B, A, seed=4)
    kw = dict(obs_radius=4, collision_system="soft", on_target="finish", max_episode_steps=10, auto_reset=True)
    ref = oracle_rollout(obstacles, agents, targets, actions, **kw)
    gc = GridConfig(map=obstacles[0].tolist(), num_agents=A, obs_radius=4, collision_system="soft", max_episode_steps=10)
    env = VecPogema(gc, batch=B, device=DEV, auto_reset=True)
    obs0 = env.reset_from_state(obstacles, agents, targets)
    got = env.rollout(torch.as_tensor(actions, device=DEV))
    out = {"obs0": obs0.cpu().numpy(), "obs": got["obs"].cpu().numpy(), "rewards": got["rewards"].cpu().numpy(),
           "terminated": got["terminated"].cpu().numpy().astype(bool), "truncated": got["truncated"].cpu().numpy().astype(bool),
           "is_active": got["is_active"].cpu().numpy().astype(bool),
           "episode_done": got["episode_done"].cpu().numpy().astype(bool)}
    ref = {k: ref[k] for k in out}
    assert_rollouts_equal(ref, out, "rollout vs oracle")


def test_rollout_argument_errors():
    gc = GridConfig(size=8, num_agents=2, obs_radius=2, seed=0)
    env = VecPogema(gc, batch=4, device=DEV)
    with pytest.raises(Exception):
        env.rollout(torch.zeros((3, 4, 2), dtype=torch.int8, device=DEV))  # before reset
    env.reset(seed=0)
    with pytest.raises(ValueError):
        env.rollout(torch.zeros((3, 4, 3), dtype=torch.int8, device=DEV))
    with pytest.raises(ValueError):
        env.rollout(torch.zeros((0, 4, 2), dtype=torch.int8, device=DEV))
    reg = VecPogema(gc, batch=4, device=DEV, auto_reset="regenerate")
    reg.reset(seed=0)
    with pytest.raises(NotImplementedError):
        reg.rollout(torch.zeros((3, 4, 2), dtype=torch.int8, device=DEV))


def test_rollout_full_size_headline_shape():
    """BASELINE configs[2] geometry, 6 steps, two observation slots: equals 6 single steps."""
    gc = GridConfig(size=64, num_agents=64, obs_radius=5, density=0.3, collision_system="soft", max_episode_steps=4, seed=0)
    a, b, ref, got = _pair(gc, 8192, 6, obs_slots=2, auto_reset=True)
    _same(ref, got, 6, obs_slots=2)
    _same_state(a, b)


def test_random_policy_rollout():
    """actions=None: the recorded actions are the stated hash, the rollout equals a rollout (and a step loop) fed with
    them, the stream does not depend on how the batch is sharded, and policy_step0 continues it."""
    from oracle.generator_oracle import policy_actions
    B, A, K = 12, 8, 20
    gc = GridConfig(size=16, num_agents=A, obs_radius=3, density=0.2, max_episode_steps=8, seed=4, collision_system="soft")
    a = VecPogema(gc, batch=B, device=DEV, auto_reset=True, env_index_base=100)
    b = VecPogema(gc, batch=B, device=DEV, auto_reset=True, env_index_base=100)
    a.reset(seed=1)
    b.reset(seed=1)
    got = a.rollout(steps=K, policy_seed=77, policy_step0=5)
    acts = got["actions"]
    np.testing.assert_array_equal(acts.cpu().numpy(), policy_actions(77, 100, B, A, 5, K))
    assert set(np.unique(acts.cpu().numpy())) == {0, 1, 2, 3, 4}
    ref = b.rollout(acts)
    for k in ("obs", "rewards", "terminated", "truncated", "is_active", "episode_done", "metrics"):
        assert torch.equal(ref[k], got[k]), k
    _same_state(a, b)
    # second half of the batch as its own shard: same actions for the same global envs; continuation by policy_step0
    c = VecPogema(gc, batch=B // 2, device=DEV, auto_reset=True, env_index_base=100 + B // 2)
    c.reset(seed=1)
    part = c.rollout(steps=K, policy_seed=77, policy_step0=5, obs_slots=0)
    assert torch.equal(part["actions"], acts[:, B // 2:])
    assert torch.equal(part["rewards"], got["rewards"][:, B // 2:])
    more = a.rollout(steps=3, policy_seed=77, policy_step0=5 + K, obs_slots=1)
    np.testing.assert_array_equal(more["actions"].cpu().numpy(), policy_actions(77, 100, B, A, 5 + K, 3))
    with pytest.raises(ValueError):
        a.rollout()


PAIR_GEOMS = [(8, 2, 3, 33), (16, 8, 5, 40), (32, 16, 5, 24), (20, 27, 4, 9), (12, 1, 2, 70), (14, 4, 7, 17)]  # G = 2, 8, 16, 32, 1, 4


@pytest.mark.parametrize("size,agents,r,batch", PAIR_GEOMS)
@pytest.mark.parametrize("collision,on_target", [("soft", "finish"), ("priority", "restart"), ("block_both", "nothing")])
def test_rollout_resolver_streamer_pair(size, agents, r, batch, collision, on_target, monkeypatch):
    """Round 6: small single-wave environments run their rollout as a PAIR of waves -- the resolver takes step t through the
    state phase while the streamer writes step t - 1's observations (step_body, PC).  PGX_ROLL_PC=1 forces the pair for every
    lane layout that has the instance; launches of 1, 2, 9 and 21 steps, rings of 1, 2, 5 and all slots: identical with the
    step() loop, outputs and state."""
    monkeypatch.setenv("PGX_ROLL_PC", "1")
    monkeypatch.setenv("PGX_WAVES", "1")  # (no helper waves for small batches of 23 KB environments: they are another kernel)
    gc = GridConfig(size=size, num_agents=agents, obs_radius=r, density=0.2, collision_system=collision, on_target=on_target,
                    max_episode_steps=7, seed=3)
    for steps, slots in ((21, None), (9, 2), (2, 5), (1, 1), (17, 1)):
        a, b, ref, got = _pair(gc, batch, steps, obs_slots=slots, auto_reset=True)
        assert b.geometry(for_rollout=True)["waves"] == 2, "the pair"
        _same(ref, got, steps, obs_slots=slots)
        _same_state(a, b)
        a.close()
        b.close()


def test_rollout_pair_is_chosen_for_small_launches_only():
    small = VecPogema(GridConfig(size=16, num_agents=8, obs_radius=5), batch=1024, device=DEV)    # BASELINE configs[1]
    large = VecPogema(GridConfig(size=32, num_agents=16, obs_radius=5), batch=8192, device=DEV)   # configs[3] shard
    assert small.geometry(for_rollout=True)["waves"] == 2 and small.geometry()["waves"] == 1
    assert large.geometry(for_rollout=True)["waves"] == 1 and large.geometry(for_rollout=True)["envs_per_wave"] == 2
    small.close()
    large.close()
