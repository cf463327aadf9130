"""GPU: the reference-style calling conventions on top of the engine -- `pogema_v0` list-per-agent
API, metrics in infos[0], POMAPF / MAPF dict observations, PettingZoo-style dict view -- against
the oracle driven with the same instance and actions."""
import numpy as np
import pytest

from oracle.pogema_oracle import PogemaOracle

pytestmark = pytest.mark.gpu


def _oracle_for(env, **kw):
    obst, agents, targets = (t.cpu().numpy() for t in env._vec._initial)
    gc = env.grid_config
    return PogemaOracle(obst[0], agents[0], targets[0], obs_radius=gc.obs_radius, collision_system=gc.collision_system,
                        on_target=gc.on_target, max_episode_steps=gc.max_episode_steps, seed=gc.seed or 0, **kw)


@pytest.mark.parametrize("on_target", ["finish", "restart", "nothing"])
def test_list_api_matches_oracle(on_target):
    from pogema_amd import GridConfig, pogema_v0
    gc = GridConfig(size=8, num_agents=2, obs_radius=3, density=0.3, seed=3, on_target=on_target, max_episode_steps=12,
                    collision_system="soft")  # BASELINE.json configs[0]
    env = pogema_v0(gc)
    obs, infos = env.reset(seed=3)
    ref = _oracle_for(env)
    assert env.get_num_agents() == 2 and env.action_space.n == 5 and env.observation_space.shape == (3, 7, 7)
    assert isinstance(obs, list) and len(obs) == 2 and obs[0].dtype == np.float32 and obs[0].shape == (3, 7, 7)
    assert all(np.array_equal(a, b) for a, b in zip(obs, ref._obs()))
    assert infos == [{"is_active": True}, {"is_active": True}]
    rng = np.random.default_rng(0)
    saw_metrics = False
    for t in range(12):
        acts = [int(a) for a in rng.integers(0, 5, size=2)]
        obs, rew, term, trunc, infos = env.step(acts)
        robs, rrew, rterm, rtrunc, rinfos = ref.step(acts)
        assert all(np.array_equal(a, b) for a, b in zip(obs, robs))
        assert rew == rrew and term == rterm and trunc == rtrunc
        assert all(isinstance(v, float) for v in rew) and all(isinstance(v, bool) for v in term + trunc)
        assert [i["is_active"] for i in infos] == [i["is_active"] for i in rinfos]
        assert env.get_agents_xy() == [tuple(p) for p in ref.get_state()["agents_xy"].tolist()]
        if "metrics" in rinfos[0]:
            saw_metrics = True
            want = rinfos[0]["metrics"]
            keys = ["avg_throughput"] if on_target == "restart" else ["ISR", "CSR", "ep_length", "SoC", "makespan"]
            assert set(infos[0]["metrics"]) == set(keys)
            for k in keys:
                assert abs(infos[0]["metrics"][k] - want[k]) < 1e-6, k
        else:
            assert "metrics" not in infos[0]
    assert saw_metrics, "the time limit ends the episode within 12 steps"
    env.close()


@pytest.mark.parametrize("obs_type", ["POMAPF", "MAPF"])
def test_dict_observations(obs_type):
    from pogema_amd import GridConfig, pogema_v0
    gc = GridConfig(size=10, num_agents=5, obs_radius=2, density=0.2, seed=1, observation_type=obs_type,
                    collision_system="priority")
    env = pogema_v0(gc)
    obs, _ = env.reset(seed=1)
    ref = _oracle_for(env)
    rng = np.random.default_rng(2)
    for t in range(10):
        want = ref.pomapf_obs(global_info=(obs_type == "MAPF"))
        assert len(obs) == 5
        for got, exp in zip(obs, want):
            assert set(got) == set(exp)
            for k in exp:
                if isinstance(exp[k], tuple):
                    assert got[k] == tuple(int(v) for v in exp[k]), k
                else:
                    assert got[k].dtype == np.float32 and np.array_equal(got[k], exp[k]), k
        acts = [int(a) for a in rng.integers(0, 5, size=5)]
        obs, *_ = env.step(acts)
        ref.step(acts)
    env.close()


def test_parallel_dict_view():
    from pogema_amd import GridConfig, pogema_v0
    env = pogema_v0(GridConfig(size=8, num_agents=3, obs_radius=2, seed=0, integration="PettingZoo", max_episode_steps=4))
    obs, infos = env.reset(seed=0)
    assert list(obs) == ["player_0", "player_1", "player_2"] == env.possible_agents
    for _ in range(4):
        obs, rew, term, trunc, infos = env.step({a: 0 for a in env.possible_agents})
    assert all(trunc.values()) and env.agents == []


def test_vec_metrics_tensor():
    """Batched metrics: rows refresh exactly where episode_done is set."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    env = VecPogema(GridConfig(size=8, num_agents=4, obs_radius=2, density=0.1, seed=0, max_episode_steps=5), batch=64,
                    auto_reset=True)
    env.reset(seed=0)
    for t in range(5):
        _, _, term, trunc, infos = env.step(torch.randint(0, 5, (64, 4), device="cuda"))
        done = infos["episode_done"]
        assert torch.equal(done, term.all(dim=1) | trunc.all(dim=1))
    assert done.all()  # time limit
    m = infos["metrics"]
    assert ((m[:, 0] >= 0) & (m[:, 0] <= 1)).all() and ((m[:, 1] == 0) | (m[:, 1] == 1)).all()
    assert (m[:, 2] >= 1).all() and (m[:, 2] <= 5).all() and (m[:, 4] <= 5).all()
    env.close()


def test_step_in_hip_graph():
    """pgx_step is one kernel launch on the caller's stream (no allocation, attribute call or sync), so it can be
    captured in a HIP graph and replayed: graph replays must equal eager stepping of a twin env."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    B, S, A = 64, 16, 8
    gc = GridConfig(size=S, num_agents=A, obs_radius=5, density=0.3, seed=4, collision_system="soft",
                    max_episode_steps=16)
    eager = VecPogema(gc, batch=B, auto_reset=True)
    graphed = VecPogema(gc, batch=B, auto_reset=True)
    o0, _ = eager.reset(seed=4)
    o1, _ = graphed.reset(seed=4)
    assert torch.equal(o0, o1)
    static_actions = torch.zeros((B, A), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on a side stream, as torch's graph recipe asks
        graphed.step(static_actions)
    torch.cuda.current_stream().wait_stream(side)
    eager.step(static_actions)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = graphed.step(static_actions)
    # the capture itself executes nothing: both envs are still one step in
    gen = torch.Generator(device="cuda").manual_seed(0)
    for t in range(40):
        acts = torch.randint(0, 5, (B, A), generator=gen, device="cuda")
        static_actions.copy_(acts)
        g.replay()
        ref = eager.step(acts)
        for a, b in zip(out[:4], ref[:4]):
            assert torch.equal(a, b), f"step {t}"
        assert torch.equal(out[4]["is_active"], ref[4]["is_active"])
    se, sg = eager.get_state(), graphed.get_state()
    for k in se:
        assert torch.equal(se[k], sg[k])
    eager.close()
    graphed.close()


def test_list_api_state_accessors():
    """`Grid`-style accessors of the single-env view after a seeded device reset and a few steps."""
    from oracle import generator_oracle as G
    from pogema_amd import GridConfig, pogema_v0
    gc = GridConfig(size=10, num_agents=4, obs_radius=2, density=0.2, seed=13, collision_system="priority")
    env = pogema_v0(gc)
    env.reset(seed=13)
    ro, ra, rt = G.generate_instance(13, 0, 10, 10, 4, 0.2)
    assert np.array_equal(env.get_obstacles(), ro)
    assert env.get_agents_xy() == [tuple(p) for p in ra.tolist()] and env.get_targets_xy() == [tuple(p) for p in rt.tolist()]
    ref = PogemaOracle(ro, ra, rt, obs_radius=2, collision_system="priority", max_episode_steps=64)
    for acts in ([1, 2, 3, 4], [4, 4, 0, 1], [2, 2, 2, 2]):
        env.step(acts)
        ref.step(acts)
    st = ref.get_state()
    assert env.get_agents_xy() == [tuple(p) for p in st["agents_xy"].tolist()]
    rel = [(int(x - sx), int(y - sy)) for (x, y), (sx, sy) in zip(st["agents_xy"], ra)]
    assert env.get_agents_xy_relative() == rel
    full = env.get_state()
    assert full["elapsed"] == 3 and full["is_active"] == [bool(v) for v in st["is_active"]]
    assert env.get_targets_xy_relative() == [(int(x - sx), int(y - sy)) for (x, y), (sx, sy) in zip(rt, ra)]
    env.close()


def test_integration_views():
    """`GridConfig.integration` dispatch: single-agent gymnasium signature, SampleFactory-style auto-reset view."""
    from pogema_amd import GridConfig, pogema_v0
    env = pogema_v0(GridConfig(size=8, num_agents=1, obs_radius=2, density=0.2, seed=5, integration="gymnasium",
                               max_episode_steps=4))
    obs, info = env.reset(seed=5)
    assert obs.shape == (3, 5, 5) and info == {"is_active": True}
    trunc = False
    for _ in range(4):
        obs, rew, term, trunc, info = env.step(0)
        assert isinstance(rew, float) and isinstance(term, bool)
    assert trunc is True and "metrics" in info
    env.close()
    sf = pogema_v0(GridConfig(size=8, num_agents=3, obs_radius=2, density=0.2, seed=5, integration="SampleFactory",
                              max_episode_steps=3))
    assert sf.is_multiagent and sf.num_agents == 3
    first, _ = sf.reset(seed=5)
    for t in range(3):
        obs, rew, term, trunc, infos = sf.step([0, 0, 0])
    assert all(trunc) and "episode_extra_stats" in infos[0]
    # auto-reset happened inside step(): the observation is the first observation of the (fixed-seed) instance
    assert all(np.array_equal(a, b) for a, b in zip(obs, first))
    sf.close()
    with pytest.raises(NotImplementedError):
        pogema_v0(GridConfig(num_agents=2, integration="PyMARL"))


def test_possible_positions_reset():
    """GridConfig.possible_agents_xy / possible_targets_xy: starts and targets come from the given cells only."""
    from oracle import generator_oracle as G
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(map="@.#$\n!..#\n.$@.\n..!.", num_agents=3, obs_radius=2, seed=8)
    env = VecPogema(gc, batch=5, env_index_base=2)
    env.reset(seed=8)
    st = env.get_state()
    for b in range(5):
        ra, rt = G.place_from_possible(8, 2 + b, gc.possible_agents_xy, gc.possible_targets_xy, 3)
        assert np.array_equal(st["agents_xy"][b].cpu().numpy(), ra) and np.array_equal(st["targets_xy"][b].cpu().numpy(), rt)
    env.close()


@pytest.mark.parametrize("on_target,auto_reset", [("finish", True), ("restart", False), ("finish", "regenerate")])
def test_snapshot_restore_continues_bit_identically(on_target, auto_reset):
    """save_state / load_state (pgx_save_snapshot / pgx_load_snapshot): checkpoint-resume and `step_back`."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    B, S, A = 10, 12, 6
    gc = GridConfig(size=S, num_agents=A, obs_radius=3, density=0.2, seed=17, on_target=on_target, max_episode_steps=5,
                    collision_system="soft")
    env = VecPogema(gc, batch=B, auto_reset=auto_reset)
    env.reset(seed=17)
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = [torch.randint(0, 5, (B, A), generator=gen, device="cuda") for _ in range(14)]
    for a in acts[:4]:
        env.step(a)
    snap = env.save_state()
    first = [tuple(t.clone() for t in env.step(a)[:4]) for a in acts[4:]]
    end_state = {k: v.clone() for k, v in env.get_state().items()}
    # (1) step back in the same handle
    env.load_state(snap)
    again = [tuple(t.clone() for t in env.step(a)[:4]) for a in acts[4:]]
    # (2) resume in a fresh handle (a checkpoint that went through host memory)
    other = VecPogema(gc, batch=B, auto_reset=auto_reset)
    other.load_state({"engine": snap["engine"].cpu(), "initial": snap["initial"], "reset_seed": snap["reset_seed"]})
    resumed = [tuple(t.clone() for t in other.step(a)[:4]) for a in acts[4:]]
    for ref, b, c in zip(first, again, resumed):
        for x, y, z in zip(ref, b, c):
            assert torch.equal(x, y) and torch.equal(x, z)
    for k, v in other.get_state().items():
        assert torch.equal(v, end_state[k])
    env.close()
    other.close()


def test_persistent_step_back():
    """GridConfig(persistent=True): `step_back()` undoes steps one by one (engine snapshots)."""
    from pogema_amd import GridConfig, pogema_v0
    env = pogema_v0(GridConfig(size=8, num_agents=3, obs_radius=2, density=0.1, seed=2, persistent=True))
    env.reset(seed=2)
    trail = [env.get_agents_xy()]
    for acts in ([1, 2, 3], [4, 4, 4], [2, 1, 0]):
        env.step(acts)
        trail.append(env.get_agents_xy())
    assert env.step_back() and env.get_agents_xy() == trail[2]
    assert env.step_back() and env.get_agents_xy() == trail[1]
    env.step([4, 4, 4])
    assert env.get_agents_xy() == trail[2]
    assert env.step_back() and env.step_back() and env.get_agents_xy() == trail[0]
    assert env.step_back() is False
    env.close()


def test_text_render():
    from pogema_amd import GridConfig, pogema_v0
    env = pogema_v0(GridConfig(map="a.#\n..B\nbA.", obs_radius=1))
    env.reset()
    assert env.render() == "a.#\n..B\nbA."
    env.step([2, 0])  # agent a moves down
    assert env.render() == "..#\na.B\nbA."
    env.close()


def test_snapshot_refuses_a_different_configuration():
    """The snapshot header records the geometry and the modes: a blob of the SAME byte size but another configuration
    (H x W swapped; another collision system; another time limit) is refused instead of installing garbage."""
    from pogema_amd import GridConfig, VecPogema
    from pogema_amd._lib import PgxError
    base = dict(num_agents=3, obs_radius=2, seed=1, max_episode_steps=9, collision_system="soft")
    tall = [[0] * 6 for _ in range(10)]
    wide = [[0] * 10 for _ in range(6)]
    a = VecPogema(GridConfig(map=tall, **base), batch=4)
    a.reset(seed=1)
    snap = a.save_state()
    for other_gc in (GridConfig(map=wide, **base), GridConfig(map=tall, **{**base, "collision_system": "priority"}),
                     GridConfig(map=tall, **{**base, "max_episode_steps": 10})):
        b = VecPogema(other_gc, batch=4)
        b.reset(seed=1)
        with pytest.raises((PgxError, ValueError), match="configuration"):
            b.load_state(snap)
        b.close()
    same = VecPogema(GridConfig(map=tall, **base), batch=4)
    same.load_state(snap)  # the matching configuration still loads
    same.close()
    a.close()


def test_reset_validates_user_supplied_cells():
    """a GridConfig with explicit `map` + `agents_xy`/`targets_xy` must be validated: an obstacle under a
    start or target is freed with a warning (upstream Grid.__init__, as recalled), two agents on one start cell raise."""
    import warnings
    from pogema_amd import GridConfig, VecPogema
    grid = "..#.\n....\n.#..\n...."
    env = VecPogema(GridConfig(map=grid, agents_xy=[[0, 2], [1, 1]], targets_xy=[[2, 1], [3, 3]], obs_radius=2), batch=3)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        obs, _ = env.reset()
    assert any("obstacle" in str(w.message) for w in seen)
    maps = env._initial[0].cpu().numpy()
    assert maps[:, 0, 2].sum() == 0 and maps[:, 2, 1].sum() == 0, "the cells under the start and the target were freed"
    st = env.get_state()
    assert st["agents_xy"][0].tolist() == [[0, 2], [1, 1]]
    assert float(obs[0, 0, 0, 2, 2]) == 0.0, "agent 0 does not stand inside a wall in its own observation"
    env.close()
    dup = VecPogema(GridConfig(map=grid, agents_xy=[[1, 1], [1, 1]], targets_xy=[[0, 0], [3, 3]], obs_radius=2), batch=2)
    with pytest.raises(KeyError, match="share a start"):
        dup.reset()
    dup.close()
    # the low-level entry keeps raising by default
    raw = VecPogema(GridConfig(map=grid, num_agents=2, obs_radius=2), batch=1)
    with pytest.raises(KeyError, match="obstacle"):
        raw.reset_from_state(np.array([[0, 0, 1, 0], [0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]], np.uint8),
                             np.array([[0, 2], [1, 1]]), np.array([[3, 0], [3, 3]]))
    raw.close()


def test_step_out_buffers_and_seed_resolved_once():
    """step(out=...) writes the caller's tensors; reset(seed=None) with GridConfig.seed=None stores the seed that was
    actually used (_resolve_seed used to be called twice)."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=10, num_agents=5, obs_radius=2, density=0.2, seed=None)
    env = VecPogema(gc, batch=6)
    env.reset()
    used = env._reset_seed
    host = env.generate(seed=used)
    assert np.array_equal(env._initial[0].cpu().numpy(), host[0]), "generate(reset_seed) reproduces the device instances"
    acts = torch.randint(0, 5, (6, 5), device="cuda")
    twin = VecPogema(gc, batch=6)
    twin.reset(seed=used)
    out = (torch.empty(env.obs_shape, device="cuda"), torch.empty((6, 5), device="cuda"),
           torch.empty((6, 5), dtype=torch.bool, device="cuda"), torch.empty((6, 5), dtype=torch.uint8, device="cuda"),
           torch.empty((6, 5), dtype=torch.bool, device="cuda"))
    o, r, te, tr, infos = env.step(acts, out=out)
    assert o.data_ptr() == out[0].data_ptr() and r.data_ptr() == out[1].data_ptr() and infos["is_active"].data_ptr() == out[4].data_ptr()
    o2, r2, te2, tr2, _ = twin.step(acts)
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(te, te2) and torch.equal(tr.bool(), tr2)
    with pytest.raises(ValueError):
        env.step(acts, out=(out[0][:3],) + out[1:])
    env.close()
    twin.close()


def test_two_handles_share_a_kernel_with_different_lds_needs():
    """the > 48 KB LDS opt-in is per kernel function, not per handle -- a later, smaller handle must not lower
    it under an earlier, larger one (both run step_kernel<64, true, true>)."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    big = VecPogema(GridConfig(size=460, num_agents=70, obs_radius=3, density=0.05, seed=1), batch=2)   # ~117 KB of LDS
    big.reset(seed=1)
    small = VecPogema(GridConfig(size=330, num_agents=70, obs_radius=3, density=0.05, seed=1), batch=2)  # ~62 KB
    small.reset(seed=1)
    a = torch.randint(0, 5, (2, 70), device="cuda")
    for _ in range(3):
        big.step(a)
        small.step(a)
    torch.cuda.synchronize()
    big.close()
    small.close()


def test_single_output_buffer_mode():
    """reuse_buffers='single': every step writes the same output set (valid until the next step); results identical."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=12, num_agents=7, obs_radius=3, density=0.2, seed=4, collision_system="soft", max_episode_steps=6)
    a = VecPogema(gc, batch=9, auto_reset=True)
    b = VecPogema(gc, batch=9, auto_reset=True, reuse_buffers="single")
    a.reset(seed=4)
    b.reset(seed=4)
    ptrs = set()
    for t in range(10):
        acts = torch.randint(0, 5, (9, 7), device="cuda", dtype=torch.int8)
        oa, ra, ta, tra, _ = a.step(acts)
        ob, rb, tb, trb, _ = b.step(acts)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(tra, trb)
        ptrs.add((ob.data_ptr(), rb.data_ptr()))
    assert len(ptrs) == 1
    with pytest.raises(ValueError):
        VecPogema(gc, batch=2, reuse_buffers="double")
    a.close()
    b.close()


def test_unequal_xcd_shares_change_nothing_but_the_schedule(monkeypatch):
    """Workgroups per XCD (pgx_xcd_shares / pgx_xcd_tune / PGX_XCD_WEIGHTS): whatever the shares, every slice is
    processed exactly once and the results are those of equal shares."""
    import ctypes as C
    import torch
    from pogema_amd import GridConfig, VecPogema, _lib as L
    from util import random_actions
    B, A = 203, 8  # 203 workgroups: shares that do not divide
    gc = GridConfig(size=16, num_agents=A, obs_radius=4, density=0.2, collision_system="soft", max_episode_steps=9, seed=3)
    actions = torch.as_tensor(random_actions(12, B, A, seed=1), device="cuda:0")

    def run(weights):
        if weights is None:
            monkeypatch.delenv("PGX_XCD_WEIGHTS", raising=False)
        else:
            monkeypatch.setenv("PGX_XCD_WEIGHTS", weights)
        env = VecPogema(gc, batch=B, device="cuda:0", auto_reset=True)
        obs0, _ = env.reset(seed=5)
        shares = (C.c_int32 * 8)()
        L.check(env._lib.pgx_xcd_shares(env._handle, shares))
        outs = [obs0.clone()]
        for t in range(12):
            o, r, te, tr, _ = env.step(actions[t])
            outs += [o.clone(), r.clone(), te.clone(), tr.clone()]
        roll = env.rollout(actions[:5])
        outs += [roll["obs"], roll["rewards"]]
        return list(shares), outs, env

    s_eq, ref, _ = run(None)
    assert sum(s_eq) == B and max(s_eq) - min(s_eq) <= 1
    for spec in ("3,1,2,1,1,1,1,1", "1,0,0,0,0,0,0,0", "0,0,0,5,0,0,0,1"):
        s, got, _ = run(spec)
        assert sum(s) == B and s != s_eq
        for a, b in zip(ref, got):
            assert torch.equal(a, b), spec
    # the tuner keeps equal shares when nothing is faster, always returns a valid partition, and leaves the state alone
    monkeypatch.delenv("PGX_XCD_WEIGHTS", raising=False)
    env = VecPogema(gc, batch=B, device="cuda:0", auto_reset=True)
    obs0, _ = env.reset(seed=5)
    buf = torch.empty_like(obs0)
    info = env.tune_xcd_shares(buf, rounds=3)
    assert sum(info["xcd_shares"]) == B and info["observe_us_tuned_shares"] <= info["observe_us_equal_shares"]
    assert torch.equal(buf, obs0)
    o, *_ = env.step(actions[0])
    assert torch.equal(o, ref[1])


def test_engine_on_a_device_that_is_not_one_spx_partition(monkeypatch):
    """the XCD-contiguous workgroup mapping, the per-XCD shares and the cohort stagger assume ONE
    compute partition of 8 XCDs.  pgx_create checks the device (256 CUs) and otherwise falls back to the identity mapping
    with equal shares and no stagger; PGX_ASSUME_PARTITIONED=1 forces that path here: same results, tuning is a no-op."""
    import numpy as np
    import torch
    from pogema_amd import GridConfig, VecPogema
    from util import assert_rollouts_equal, c_oracle_rollout, engine_rollout, generate_instances, random_actions
    assert VecPogema(GridConfig(num_agents=2), batch=2).geometry()["xcd_aware"] == 1, "this pool's MI355X run SPX"
    monkeypatch.setenv("PGX_ASSUME_PARTITIONED", "1")
    for name, B, S, A, r in (("small_groups", 300, 16, 8, 5), ("full_wave", 40, 24, 64, 5), ("four_waves", 6, 40, 200, 7)):
        obstacles, agents, targets = generate_instances(B, S, S, A, 0.2, 17)
        actions = random_actions(10, B, A, 3)
        kw = dict(obs_radius=r, collision_system="soft", on_target="restart", max_episode_steps=6, auto_reset=True, seed=4)
        assert_rollouts_equal(c_oracle_rollout(obstacles, agents, targets, actions, nthreads=4, **kw),
                              engine_rollout(obstacles, agents, targets, actions, **kw), f"partitioned/{name}")
    # a launch that would get the cohort stagger and tuned shares on an SPX device
    env = VecPogema(GridConfig(size=32, num_agents=32, obs_radius=5, density=0.2, seed=1), batch=8192, auto_reset=True,
                    reuse_buffers=False)
    obs, _ = env.reset(seed=1)
    g = env.geometry()
    assert g["xcd_aware"] == 0 and g["stagger"] == 0 and g["grid"] == -(-8192 // g["envs_per_wave"])
    tuned = env.tune_xcd_shares(obs)
    assert tuned["observe_us_equal_shares"] == 0.0 and tuned["xcd_shares"] == [g["grid"] // 8] * 8
    monkeypatch.delenv("PGX_ASSUME_PARTITIONED")
    ref = VecPogema(GridConfig(size=32, num_agents=32, obs_radius=5, density=0.2, seed=1), batch=8192, auto_reset=True,
                    reuse_buffers=False)
    robs, _ = ref.reset(seed=1)
    assert ref.geometry()["xcd_aware"] == 1 and torch.equal(obs, robs)
    acts = torch.randint(0, 5, (4, 8192, 32), device="cuda", dtype=torch.int8)
    for t in range(4):
        a, b = env.step(acts[t]), ref.step(acts[t])
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    env.close(); ref.close()


@pytest.mark.parametrize("fmt", ["bfloat16", "float16", "uint8"])
def test_light_formats_on_every_observation_path(fmt):
    """The non-drop-in observation formats go through every call that writes observations, not only step(): observe(),
    observe(out=), step(out=), the masked observe of auto_reset='regenerate', reset_where -- each equal to the float32
    engine's tensor cast to the format."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    dtype = getattr(torch, fmt)
    gc = GridConfig(size=20, num_agents=24, obs_radius=4, density=0.25, seed=3, collision_system="soft", max_episode_steps=5)
    B = 300
    a = VecPogema(gc, batch=B, auto_reset="regenerate", obs_dtype=dtype)
    b = VecPogema(gc, batch=B, auto_reset="regenerate")
    oa, _ = a.reset(seed=11)
    ob, _ = b.reset(seed=11)
    assert oa.dtype == dtype and torch.equal(oa.float(), ob)
    mine = (torch.empty(a.obs_shape, dtype=dtype, device="cuda"), torch.empty((B, 24), device="cuda"),
            torch.empty((B, 24), dtype=torch.bool, device="cuda"), torch.empty((B, 24), dtype=torch.bool, device="cuda"),
            torch.empty((B, 24), dtype=torch.bool, device="cuda"))
    gen = torch.Generator(device="cuda").manual_seed(2)
    for t in range(12):  # max_episode_steps 5: two rounds of on-device regeneration with rewritten observations
        acts = torch.randint(0, 5, (B, 24), generator=gen, device="cuda", dtype=torch.int8)
        xa = a.step(acts, out=mine) if t % 2 else a.step(acts)
        xb = b.step(acts)
        assert xa[0].dtype == dtype and torch.equal(xa[0].float(), xb[0]), f"step {t}"
        assert torch.equal(xa[1], xb[1]) and torch.equal(xa[4]["episode_done"], xb[4]["episode_done"])
    assert torch.equal(a.observe().float(), b.observe())
    scratch = torch.zeros(a.obs_shape, dtype=dtype, device="cuda")
    assert a.observe(out=scratch) is scratch and torch.equal(scratch.float(), b.observe())
    mask = torch.arange(B, device="cuda") % 3 == 0
    assert torch.equal(a.reset_where(mask, seed=5).float(), b.reset_where(mask, seed=5))
    with pytest.raises(ValueError):
        a.observe(out=torch.zeros(a.obs_shape, device="cuda"))  # float32 buffer for a bfloat16 engine
    a.close(); b.close()


def test_limits_are_value_errors_that_name_the_limit():
    """README "Limits": configurations GridConfig admits and the engine cannot hold come back as ValueError from
    VecPogema.__init__ (never a late engine error): radius / agents by their range, the LDS budget by its byte count."""
    from pogema_amd import GridConfig, VecPogema
    with pytest.raises(ValueError, match="obs_radius=16"):
        VecPogema(GridConfig(size=32, num_agents=2, obs_radius=16), batch=1)
    with pytest.raises(ValueError, match="bytes of LDS"):
        VecPogema(GridConfig(size=64, num_agents=1024, obs_radius=8, density=0.0), batch=1)
    env = VecPogema(GridConfig(size=1024, num_agents=64, obs_radius=5, density=0.3, seed=1), batch=2)  # PGX_MAX_SIDE: accepted
    obs, _ = env.reset(seed=1)
    assert env.geometry()["multi_wave"] == 2 and tuple(obs.shape) == (2, 64, 3, 11, 11)
    env.close()
