"""The one-thread-per-env generator kernel (pgx_np_generate) against the numpy restatement, and VecPogema with
Semantics(generator_rng='numpy') end to end against the oracle env started from the numpy-generated instance."""
import numpy as np
import pytest
import torch

from oracle import generator_oracle as G
from pogema_amd import GridConfig, Semantics, VecPogema
from pogema_amd.nprng import np_generate, np_generate_host
from util import oracle_rollout, random_actions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,A,density", [(8, 8, 2, 0.3), (32, 32, 16, 0.3), (64, 64, 64, 0.3), (17, 9, 20, 0.15)])
def test_device_generator_equals_host_and_numpy(H, W, A, density):
    seeds = np.concatenate([np.arange(200, dtype=np.uint64), np.array([2 ** 40 + 3, 2 ** 63 + 11], np.uint64)])
    o, a, t, st = (v.cpu().numpy() for v in np_generate(seeds, H, W, A, density))
    ho, ha, ht, hst = np_generate_host(seeds, H, W, A, density)
    np.testing.assert_array_equal(st, hst)
    ok = st == 0
    np.testing.assert_array_equal(o, ho)
    np.testing.assert_array_equal(a[ok], ha[ok])
    np.testing.assert_array_equal(t[ok], ht[ok])
    for i in np.flatnonzero(ok)[:12]:
        ro, ra, rt = G.generate_instance_numpy(int(seeds[i]), H, W, A, density)
        np.testing.assert_array_equal(o[i], ro)
        np.testing.assert_array_equal(a[i], ra)
        np.testing.assert_array_equal(t[i], rt)


def test_large_batch_matches_host():
    seeds = np.arange(4096, dtype=np.uint64) + np.uint64(77)
    o, a, t, st = (v.cpu().numpy() for v in np_generate(seeds, 32, 32, 32, 0.3))
    ho, ha, ht, hst = np_generate_host(seeds, 32, 32, 32, 0.3)
    np.testing.assert_array_equal(st, hst)
    np.testing.assert_array_equal(o, ho)
    np.testing.assert_array_equal(a[st == 0], ha[st == 0])
    np.testing.assert_array_equal(t[st == 0], ht[st == 0])


@pytest.mark.parametrize("on_target", ["finish", "nothing", "restart"])
def test_vec_env_numpy_generator_end_to_end(on_target):
    """reset() with generator_rng='numpy' starts every env from the instance numpy draws for its seed, and the episode
    from there equals the oracle's started from that instance."""
    B, A, size, seed, base, T = 6, 8, 16, 41, 3, 24
    gc = GridConfig(num_agents=A, size=size, density=0.3, obs_radius=3, on_target=on_target, max_episode_steps=16, seed=seed)
    env = VecPogema(gc, batch=B, device="cuda:0", env_index_base=base, semantics=Semantics(generator_rng="numpy"),
                    auto_reset=True)
    obs0, _ = env.reset()
    inst = [G.generate_instance_numpy(seed + base + b, size, size, A, 0.3) for b in range(B)]
    obstacles, agents, targets = (np.stack([i[k] for i in inst]) for k in range(3))
    st = env.get_state()
    np.testing.assert_array_equal(st["agents_xy"].cpu().numpy(), agents)
    np.testing.assert_array_equal(st["targets_xy"].cpu().numpy(), targets)
    actions = random_actions(T, B, A, seed=5)
    ref = oracle_rollout(obstacles, agents, targets, actions, obs_radius=3, collision_system="priority", on_target=on_target,
                         max_episode_steps=16, auto_reset=True, seed=seed, env_index_base=base)
    np.testing.assert_array_equal(obs0.cpu().numpy(), ref["obs0"])
    d_actions = torch.as_tensor(actions, device="cuda:0")
    for t in range(T):
        obs, rew, term, trunc, infos = env.step(d_actions[t])
        np.testing.assert_array_equal(obs.cpu().numpy(), ref["obs"][t], err_msg=f"obs at step {t}")
        np.testing.assert_allclose(rew.cpu().numpy(), ref["rewards"][t], atol=1e-6)
        np.testing.assert_array_equal(term.cpu().numpy(), ref["terminated"][t])
        np.testing.assert_array_equal(trunc.cpu().numpy(), ref["truncated"][t])
    hobst, hag, htg = env.generate()
    np.testing.assert_array_equal(hobst, obstacles)
    np.testing.assert_array_equal(hag, agents)
    np.testing.assert_array_equal(htg, targets)


def test_vec_env_numpy_generator_overflow_and_given_map():
    gc = GridConfig(num_agents=9, size=4, density=0.0, obs_radius=2, seed=1)
    env = VecPogema(gc, batch=2, device="cuda:0", semantics=Semantics(generator_rng="numpy"))
    with pytest.raises(OverflowError):
        env.reset()
    m = [[0, 0, 1, 0, 0], [0, 1, 1, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 1, 0]]
    gc = GridConfig(num_agents=3, map=m, obs_radius=2, seed=9)
    env = VecPogema(gc, batch=4, device="cuda:0", semantics=Semantics(generator_rng="numpy"))
    env.reset()
    st = env.get_state()
    for b in range(4):
        _, ra, rt = G.generate_instance_numpy(9 + b, 4, 5, 3, 0.3, np.array(m))
        np.testing.assert_array_equal(st["agents_xy"][b].cpu().numpy(), ra)
        np.testing.assert_array_equal(st["targets_xy"][b].cpu().numpy(), rt)
    with pytest.raises(NotImplementedError):
        env.reset_where(torch.ones(4, dtype=torch.bool))
