"""GPU: on-device reset (pgx_reset_random: generator GEN v2, component labelling, placement, lifelong tables)
against the CPU oracle restatements (oracle/generator_oracle.py, po_generate in oracle/pogema_oracle.c)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import generator_oracle as G
from oracle.c_oracle import load as load_c_oracle
from util import assert_rollouts_equal, oracle_rollout, random_actions

pytestmark = pytest.mark.gpu


def _c_generate(B, H, W, A, density, seed, base=0, epochs=None, given_map=None, max_retries=10):
    lib = load_c_oracle()
    lib.po_generate.argtypes = [C.c_int32] * 4 + [C.c_float, C.c_uint64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.po_generate.restype = C.c_int
    obst = np.ascontiguousarray(given_map, np.uint8) if given_map is not None else np.empty((B, H, W), np.uint8)
    a = np.empty((B, A, 2), np.int32)
    t = np.empty((B, A, 2), np.int32)
    ep = None if epochs is None else np.ascontiguousarray(epochs, np.uint32).ctypes.data
    st = lib.po_generate(B, H, W, A, density, seed, base, ep, max_retries, int(given_map is not None),
                         obst.ctypes.data, a.ctypes.data, t.ctypes.data)
    return st, obst, a, t


def _device_state(env):
    st = env.get_state()
    maps = env._initial[0].cpu().numpy()
    return maps, st["agents_xy"].cpu().numpy(), st["targets_xy"].cpu().numpy()


SMALL = [
    # name, B, H, W, A, r, density, on_target, env_index_base, seed
    ("cfg0", 7, 8, 8, 2, 3, 0.3, "finish", 0, 11),
    ("cfg1", 9, 16, 16, 8, 5, 0.3, "finish", 5, 3),
    ("dense_agents", 5, 13, 13, 30, 4, 0.1, "nothing", 100, 77),
    ("lifelong", 6, 20, 20, 12, 3, 0.35, "restart", 17, 2),
    ("many_agents", 3, 18, 18, 100, 3, 0.1, "restart", 0, 9),
]


@pytest.mark.parametrize("cfg", SMALL, ids=[c[0] for c in SMALL])
def test_device_reset_equals_python_oracle(cfg):
    from pogema_amd import GridConfig, VecPogema
    name, B, H, W, A, r, density, on_target, base, seed = cfg
    gc = GridConfig(size=H, num_agents=A, obs_radius=r, density=density, on_target=on_target, seed=seed,
                    max_episode_steps=8, collision_system="soft")
    env = VecPogema(gc, batch=B, env_index_base=base)
    obs, infos = env.reset(seed=seed)
    maps, agents, targets = _device_state(env)
    ro, ra, rt = G.generate_batch(seed, B, H, W, A, density, env_index_base=base)
    assert np.array_equal(maps, ro), f"{name}: obstacles"
    assert np.array_equal(agents, ra) and np.array_equal(targets, rt), f"{name}: placement"
    # the host generator of the product draws the same instances
    ho, ha, ht = env.generate(seed)
    assert np.array_equal(ho, ro) and np.array_equal(ha, ra) and np.array_equal(ht, rt)
    # rollout from the device-built state (incl. device-built lifelong tables) == oracle from the same state
    actions = random_actions(10, B, A, seed + 1)
    kw = dict(obs_radius=r, collision_system="soft", on_target=on_target, max_episode_steps=8, auto_reset=False,
              seed=seed, env_index_base=base)
    ref = oracle_rollout(ro, ra, rt, actions, **kw)
    assert np.array_equal(obs.cpu().numpy(), ref["obs0"])
    got = {k: np.zeros_like(v) for k, v in ref.items() if k not in ("obs0",)}
    got["occupancy0"] = env.get_state(occupancy=True)["occupancy"].cpu().numpy()
    d_act = torch.from_numpy(actions).cuda()
    for t in range(actions.shape[0]):
        o, rew, term, trunc, info = env.step(d_act[t])
        st = env.get_state(occupancy=True)
        got["occupancy"][t] = st["occupancy"].cpu().numpy()
        got["obs"][t], got["rewards"][t] = o.cpu().numpy(), rew.cpu().numpy()
        got["terminated"][t], got["truncated"][t] = term.cpu().numpy(), trunc.cpu().numpy()
        got["is_active"][t] = info["is_active"].cpu().numpy()
        got["episode_done"][t] = info["episode_done"].cpu().numpy()
        got["metrics"][t] = np.where(got["episode_done"][t][:, None], info["metrics"].cpu().numpy(), 0)
        got["agents_xy"][t], got["targets_xy"][t] = st["agents_xy"].cpu().numpy(), st["targets_xy"].cpu().numpy()
        got["elapsed"][t] = st["elapsed"].cpu().numpy()
    got["obs0"] = ref["obs0"]
    assert_rollouts_equal(ref, got, f"{name}: rollout after device reset")
    env.close()


def test_shared_map_and_masked_regeneration():
    from pogema_amd import GridConfig, VecPogema
    H, W, A, B, seed, base = 7, 9, 3, 12, 42, 30
    m = np.zeros((H, W), np.uint8)
    m[3, :] = 1  # a wall splits the map into two components
    m[3, 4] = 0  # ... with one door
    m[0, 0] = 1
    gc = GridConfig(map=m.tolist(), num_agents=A, obs_radius=2, on_target="restart", seed=seed)
    env = VecPogema(gc, batch=B, env_index_base=base)
    env.reset(seed=seed)
    maps, agents, targets = _device_state(env)
    ro, ra, rt = G.generate_batch(seed, B, H, W, A, 0.0, env_index_base=base, given_map=m)
    assert np.array_equal(maps, ro) and np.array_equal(agents, ra) and np.array_equal(targets, rt)
    # step a little, then regenerate a subset: flagged envs get generation 1, the others keep their state
    acts = torch.randint(0, 5, (B, A), device="cuda")
    for _ in range(3):
        env.step(acts)
    before = env.get_state()
    mask = np.zeros(B, bool)
    mask[[1, 4, 5, 11]] = True
    env.reset_where(torch.from_numpy(mask).cuda())
    after = env.get_state()
    for b in range(B):
        if mask[b]:
            _, ea, et = G.generate_instance(seed, base + b, H, W, A, 0.0, epoch=1, given_map=m)
            assert np.array_equal(after["agents_xy"][b].cpu().numpy(), ea)
            assert np.array_equal(after["targets_xy"][b].cpu().numpy(), et)
            assert int(after["elapsed"][b]) == 0 and bool(after["is_active"][b].all())
        else:
            for k in ("agents_xy", "targets_xy", "elapsed", "is_active"):
                assert torch.equal(after[k][b], before[k][b])
    env.close()


def test_masked_regeneration_random_maps_epoch2():
    from pogema_amd import GridConfig, VecPogema
    B, S, A, seed = 10, 12, 5, 8
    env = VecPogema(GridConfig(size=S, num_agents=A, obs_radius=3, density=0.25, seed=seed), batch=B)
    env.reset(seed=seed)
    mask = torch.zeros(B, dtype=torch.bool, device="cuda")
    mask[2] = mask[7] = True
    env.reset_where(mask)
    env.reset_where(mask)  # generation 2 for envs 2 and 7
    maps, agents, targets = _device_state(env)
    for b in range(B):
        eo, ea, et = G.generate_instance(seed, b, S, S, A, 0.25, epoch=2 if b in (2, 7) else 0)
        assert np.array_equal(maps[b], eo) and np.array_equal(agents[b], ea) and np.array_equal(targets[b], et), b
    # a full reset returns every env to generation 0
    env.reset(seed=seed)
    maps2, agents2, _ = _device_state(env)
    eo, ea, _ = G.generate_instance(seed, 2, S, S, A, 0.25, epoch=0)
    assert np.array_equal(maps2[2], eo) and np.array_equal(agents2[2], ea)
    env.close()


FULL = [("configs2", 8192, 64, 64, 5, 0.3), ("configs3_shard", 8192, 32, 16, 5, 0.3), ("configs4_part", 384, 256, 256, 7, 0.3),
        ("configs1", 1024, 16, 8, 5, 0.3),
        ("big_map", 3, 600, 20, 5, 0.25)]  # far beyond the LDS forest: union-find and component tables through L2, 360 000 cells


@pytest.mark.parametrize("cfg", FULL, ids=[c[0] for c in FULL])
@pytest.mark.parametrize("on_target", ["finish", "restart"])
def test_full_size_device_reset_equals_c_oracle(cfg, on_target):
    from pogema_amd import GridConfig, VecPogema
    name, B, S, A, r, density = cfg
    seed, base = 123, 4096
    env = VecPogema(GridConfig(size=S, num_agents=A, obs_radius=r, density=density, seed=seed, on_target=on_target),
                    batch=B, env_index_base=base)
    env.reset(seed=seed)
    maps, agents, targets = _device_state(env)
    st, ro, ra, rt = _c_generate(B, S, S, A, density, seed, base)
    assert st == 0
    assert np.array_equal(maps, ro), f"{name}: obstacles"
    assert np.array_equal(agents, ra) and np.array_equal(targets, rt), f"{name}: placement"
    env.close()


def test_unplaceable_instances_raise():
    from pogema_amd import GridConfig, VecPogema, _lib
    env = VecPogema(GridConfig(size=6, num_agents=12, obs_radius=2, density=0.9, seed=1), batch=4)
    with pytest.raises(_lib.PgxError) as ei:
        env.reset(seed=1)
    assert ei.value.code == -5
    env.close()


@pytest.mark.parametrize("on_target,empty_outside", [("finish", True), ("restart", True), ("finish", False)])
def test_regenerate_mode_tracks_oracle(on_target, empty_outside):
    """auto_reset='regenerate' (pgx_regenerate, no host sync): a finished env continues on a NEW instance whose
    generation counter advanced; everything is checked step by step against per-env Python oracles that are
    rebuilt from the generator oracle at every episode end."""
    from oracle.pogema_oracle import PogemaOracle
    from pogema_amd import GridConfig, VecPogema
    B, S, A, r, seed, base, T = 12, 10, 4, 2, 21, 7, 5
    gc = GridConfig(size=S, num_agents=A, obs_radius=r, density=0.25, seed=seed, on_target=on_target,
                    max_episode_steps=T, collision_system="priority", empty_outside=empty_outside)
    env = VecPogema(gc, batch=B, env_index_base=base, auto_reset="regenerate")
    obs, _ = env.reset(seed=seed)

    def fresh(b, epoch):
        o, a, t = G.generate_instance(seed, base + b, S, S, A, 0.25, epoch=epoch)
        return PogemaOracle(o, a, t, obs_radius=r, collision_system="priority", on_target=on_target,
                            max_episode_steps=T, auto_reset=False, seed=seed, env_index=base + b,
                            empty_outside=empty_outside, outside_density=0.25, epoch=epoch)

    epochs = [0] * B
    refs = [fresh(b, 0) for b in range(B)]
    assert np.array_equal(obs.cpu().numpy(), np.stack([np.stack(e._obs()) for e in refs]))
    rng = np.random.default_rng(5)
    for t in range(3 * T + 2):
        acts = rng.integers(0, 5, size=(B, A))
        obs, rew, term, trunc, info = env.step(torch.from_numpy(acts).cuda())
        st = env.get_state()
        for b in range(B):
            robs, rrew, rterm, rtrunc, rinfo = refs[b].step(acts[b])
            assert rew[b].tolist() == rrew and term[b].tolist() == rterm and trunc[b].tolist() == rtrunc, (t, b)
            done = all(rterm) or all(rtrunc)
            assert bool(info["episode_done"][b]) == done
            if done:
                epochs[b] += 1
                refs[b] = fresh(b, epochs[b])
                robs = refs[b]._obs()
            rs = refs[b].get_state()
            assert np.array_equal(obs[b].cpu().numpy(), np.stack(robs)), (t, b)
            assert np.array_equal(st["agents_xy"][b].cpu().numpy(), rs["agents_xy"])
            assert np.array_equal(st["targets_xy"][b].cpu().numpy(), rs["targets_xy"])
            assert int(st["elapsed"][b]) == rs["elapsed"] and st["is_active"][b].cpu().numpy().tolist() == rs["is_active"].astype(bool).tolist()
    assert max(epochs) >= 3 and env.regenerate_failures() == 0
    env.close()


def test_regenerate_failure_keeps_previous_instance():
    """When no fresh instance can be placed, the env falls back to its stored initial state (plain auto-reset)
    and the failure is counted -- nothing is left half-written."""
    from pogema_amd import GridConfig, VecPogema
    m = np.ones((6, 6), np.uint8)
    m[0, :4] = 0  # one corridor of 4 free cells: 2 agents fit exactly once ...
    gc = GridConfig(map=m.tolist(), num_agents=2, obs_radius=2, seed=3, max_episode_steps=3)
    env = VecPogema(gc, batch=5, auto_reset="regenerate")
    env.reset(seed=3)
    first = env.get_state()
    env._shared = torch.ones((6, 6), dtype=torch.uint8, device="cuda")  # ... and from now on the "map" is full
    for _ in range(3):
        env.step(torch.zeros((5, 2), dtype=torch.int64, device="cuda"))
    after = env.get_state()
    assert env.regenerate_failures() == 5
    assert torch.equal(after["agents_xy"], first["agents_xy"]) and torch.equal(after["targets_xy"], first["targets_xy"])
    assert int(after["elapsed"].max()) == 0 and bool(after["is_active"].all())
    assert np.array_equal(env._initial[0].cpu().numpy(), np.broadcast_to(m, (5, 6, 6)))
    env.close()


def test_odd_resets_equal_the_generator_oracle():
    """The reset path in the corners: maps of 2..7 cells per side (rectangular through `map=` of free cells is not random --
    so squares), 1 agent up to as many as fit, densities from 0 to 'almost full', odd env index bases, lifelong and not:
    device reset == generator oracle, or BOTH say the instance cannot be placed."""
    from pogema_amd import GridConfig, VecPogema, _lib
    rng = np.random.default_rng(77)
    placed = refused = 0
    for case in range(60):
        S = int(rng.integers(2, 8))
        A = int(rng.integers(1, max(2, S * S // 2)))
        density = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7]))
        seed, base, B = int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 1000)), int(rng.integers(1, 6))
        on_target = str(rng.choice(["finish", "restart", "nothing"]))
        what = f"case {case}: {S}x{S}, {A} agents, density {density}, seed {seed}, base {base}, batch {B}, {on_target}"
        try:
            ro, ra, rt = G.generate_batch(seed, B, S, S, A, density, env_index_base=base)
        except OverflowError:
            ro = None
        env = VecPogema(GridConfig(size=S, num_agents=A, obs_radius=int(rng.integers(1, 4)), density=density, seed=seed,
                                   on_target=on_target), batch=B, env_index_base=base)
        if ro is None:
            with pytest.raises((_lib.PgxError, OverflowError)):
                env.reset(seed=seed)
            refused += 1
        else:
            env.reset(seed=seed)
            maps, agents, targets = _device_state(env)
            assert np.array_equal(maps, ro), f"{what}: obstacles"
            assert np.array_equal(agents, ra) and np.array_equal(targets, rt), f"{what}: placement"
            placed += 1
        env.close()
    assert placed >= 20 and refused >= 3, (placed, refused)
