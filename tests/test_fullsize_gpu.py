"""GPU, BASELINE.json full sizes: (1) bit-exact parity with the plain-C oracle port for a few steps at
the full batch of configs[1], configs[2], configs[3] (one GPU's shard; the whole 65536-env batch as 8 shards:
tests/test_sharded_gpu.py) and configs[4], each on the launch geometry bench.py times; (2) size-independent
invariants over longer rollouts (no oracle needed): cell exclusivity, nobody on an obstacle,
observation-plane identities, determinism, auto-reset returns to the stored initial state."""
import numpy as np
import pytest
import torch

from util import generate_instances

pytestmark = pytest.mark.gpu


def _engine(B, size, A, r, collision, on_target, max_steps, auto_reset, obstacles, agents, targets, seed=0):
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system=collision, on_target=on_target,
                    max_episode_steps=max_steps, seed=seed, density=0.3)
    env = VecPogema(gc, batch=B, auto_reset=auto_reset)
    obs0 = env.reset_from_state(obstacles, agents, targets, validate=False)
    return env, obs0


def goal_seeking_actions(rng, agents_xy, targets_xy, p_random=0.5):
    """Uniform random moves mixed with moves along the larger coordinate difference to the target (no obstacle
    awareness), so that within a few steps agents arrive, finish, get new targets and pile up around goals."""
    d = targets_xy.astype(np.int64) - agents_xy.astype(np.int64)
    along_x = np.abs(d[..., 0]) >= np.abs(d[..., 1])
    greedy = np.where(along_x, np.where(d[..., 0] < 0, 1, 2), np.where(d[..., 1] < 0, 3, 4))
    greedy = np.where((d == 0).all(axis=-1), 0, greedy)
    rnd = rng.integers(0, 5, size=greedy.shape)
    return np.where(rng.random(greedy.shape) < p_random, rnd, greedy).astype(np.int64)


def _assert_same_step(name, t, env, ref, out, rout, check_obs):
    obs, rew, term, trunc, info = out
    robs, rrew, rterm, rtrunc, ract = rout
    st, rst = env.get_state(), ref.get_state()
    assert np.array_equal(st["agents_xy"].cpu().numpy(), rst["agents_xy"]), f"{name} step {t}: positions"
    assert np.array_equal(st["targets_xy"].cpu().numpy(), rst["targets_xy"]), f"{name} step {t}: targets"
    assert np.array_equal(st["is_active"].cpu().numpy(), rst["is_active"]), f"{name} step {t}: is_active"
    assert np.array_equal(st["elapsed"].cpu().numpy(), rst["elapsed"]), f"{name} step {t}: elapsed"
    assert np.array_equal(term.cpu().numpy(), rterm) and np.array_equal(trunc.cpu().numpy(), rtrunc), f"{name} step {t}: flags"
    assert np.array_equal(info["is_active"].cpu().numpy(), ract)
    np.testing.assert_allclose(rew.cpu().numpy(), rrew, rtol=0, atol=1e-6)
    done = info["episode_done"].cpu().numpy().astype(bool)
    assert np.array_equal(done, ref.episode_done.astype(bool)), f"{name} step {t}: episode_done"
    np.testing.assert_allclose(info["metrics"].cpu().numpy()[done], ref.metrics[done], rtol=1e-6, atol=1e-6)
    if check_obs:
        assert np.array_equal(obs.cpu().numpy(), robs), f"{name} step {t}: observations"
    return rst


_BENCH_GEOMETRY = {}


def bench_geometry(workload):
    """Launch shape of the engine bench.py times for `workload` (bench.build_env with bench.py's default arguments)."""
    if workload not in _BENCH_GEOMETRY:
        import bench
        per_gpu, size, agents, r = bench.WORKLOADS[workload]
        args = bench.make_parser().parse_args(["--workload", workload])
        env = bench.build_env(args, "cuda:0", per_gpu, 0, size, agents, r)
        _BENCH_GEOMETRY[workload] = (env.geometry(), env.geometry(for_rollout=True))
        env.close()
    return _BENCH_GEOMETRY[workload]


def assert_bench_geometry(env, workload):
    """The parity engine must run the kernel variant and launch shape the benchmark times (everything but `grid`, which
    the XCD-share tuning of an engine with warmed buffers adjusts by a few workgroups)."""
    for mine, theirs in zip((env.geometry(), env.geometry(for_rollout=True)), bench_geometry(workload)):
        mine, theirs = dict(mine), dict(theirs)
        assert abs(mine.pop("grid") - theirs.pop("grid")) <= 0.2 * theirs.get("grid", 1 << 30) + 64
        assert mine == theirs, f"{workload}: parity runs {mine}, bench.py times {theirs}"


def test_bench_geometry_is_what_design_md_says():
    """DESIGN.md section 5's table of launch shapes, as chosen by step_geometry() for bench.py's engines."""
    want = {  # workload: (lanes_per_env, envs_per_wave, waves, p16, store_policy)
        "cfg1": (8, 1, 1, 1, 2), "cfg2": (64, 1, 3, 1, 1), "cfg3": (16, 1, 1, 1, 2), "cfg4": (64, 1, 4, 1, 1)}
    for workload, shape in want.items():
        g = bench_geometry(workload)[0]
        assert (g["lanes_per_env"], g["envs_per_wave"], g["waves"], g["p16"], g["store_policy"]) == shape, (workload, g)


def test_geometry_reports_the_store_flavour_as_executed():
    """the generic funnels of the lighter observation formats know plain and nontemporal stores only; where the
    launch shape asks for sc1 (single-wave kernels) they run -- and pgx_get_geometry now reports -- plain stores.  float32
    and the 16-bit formats on the fast row walk (window >= 7 cells, 16-bit row masks) honour all three flavours."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    want = {  # (obs_radius, dtype) -> store_policy of a 16-agent single-wave launch (the shape asks for sc1 = 2)
        (5, torch.float32): 2, (5, torch.bfloat16): 2, (5, torch.float16): 2, (5, torch.uint8): 0,
        (2, torch.bfloat16): 0,   # window of 5 cells: the generic 16-bit funnel
        (9, torch.float32): 2, (9, torch.float16): 0, (9, torch.uint8): 0,   # window of 19 cells: beyond the 16-bit row masks
    }
    for (r, dt), policy in want.items():
        env = VecPogema(GridConfig(size=24, num_agents=16, obs_radius=r, density=0.2, seed=1), batch=2048, obs_dtype=dt,
                        reuse_buffers=False, placement_budget_gib=0)  # (>= 1024 envs: no helper waves; nothing is allocated)
        g = env.geometry()
        assert g["multi_wave"] == 0 and g["store_policy"] == policy, (r, dt, g)
        env.close()


FULL = [
    # name, batch, size, agents, r, steps, max_episode_steps, on_target modes
    ("configs1", 1024, 16, 8, 5, 24, 8, ("finish", "restart", "nothing")),
    ("configs2", 8192, 64, 64, 5, 16, 8, ("finish", "restart", "nothing")),
    ("configs3_shard", 8192, 32, 16, 5, 16, 8, ("finish", "restart")),
    ("configs4", 4096, 256, 256, 7, 4, 3, ("finish",)),
]
FULL_CASES = [(c, ot) for c in FULL for ot in c[7]]


@pytest.mark.parametrize("cfg,on_target", FULL_CASES, ids=[f"{c[0]}-{ot}" for c, ot in FULL_CASES])
@pytest.mark.parametrize("collision", ["soft", "priority", "block_both"])
def test_full_size_parity_with_c_oracle(cfg, on_target, collision):
    """The FULL batch of BASELINE.json configs[2] / configs[3] (one GPU's shard) / configs[4] against the plain-C oracle:
    every episode mode, max_episode_steps 8 and 16 steps (two episodes through the auto-reset) for configs[2]/[3],
    goal-seeking actions so that arrivals, finished/hidden agents, lifelong re-targets and cooperative termination
    all occur; state, flags, rewards and fused metrics every step, the full observation tensor on the first, the
    auto-reset and the last step (collision system 'soft' only: the observation path is the same for all three)."""
    from oracle.c_oracle import COracle
    name, B, size, A, r, T, max_steps, _ = cfg
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, 7)
    rng = np.random.default_rng(3)
    ref = COracle(B, size, size, A, r, collision, on_target, max_steps, True, seed=9, env_index_base=5)
    ref_obs0 = ref.reset(obstacles, agents, targets)
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system=collision, on_target=on_target,
                    max_episode_steps=max_steps, seed=9, density=0.3)
    env = VecPogema(gc, batch=B, auto_reset=True, env_index_base=5)
    assert_bench_geometry(env, {"configs1": "cfg1", "configs2": "cfg2", "configs3_shard": "cfg3", "configs4": "cfg4"}[name])
    obs0 = env.reset_from_state(obstacles, agents, targets, validate=False)
    # configs[1]: the launch is 12 MB -- observations are compared on every step, for every collision system
    obs_steps = set(range(T)) if name == "configs1" else {0, max_steps - 1, T - 1} if collision == "soft" else set()
    if obs_steps:
        assert np.array_equal(obs0.cpu().numpy(), ref_obs0)
    del ref_obs0, obs0
    threads = min(32, len(__import__("os").sched_getaffinity(0)))
    state = ref.get_state()
    events = 0
    for t in range(T):
        acts = goal_seeking_actions(rng, state["agents_xy"], state["targets_xy"])
        check_obs = t in obs_steps
        rout = ref.step(acts, nthreads=threads, compute_obs=check_obs)
        out = env.step(torch.from_numpy(acts).cuda(), compute_obs=check_obs)
        state = _assert_same_step(f"{name}/{on_target}/{collision}", t, env, ref, out, rout, check_obs)
        # rewarded arrivals; cooperative finish pays only when ALL agents of an env are home at once (practically never
        # at 64 agents), so there the agents standing on their goals -- visible, unrewarded -- are what is exercised
        events += int(rout[1].sum()) if on_target != "nothing" else int((state["agents_xy"] == state["targets_xy"]).all(-1).sum())
    assert events > 0, "the rollout must contain arrivals (otherwise the episode logic was not exercised)"
    env.close()
    ref.close()


@pytest.mark.parametrize("collision", ["soft", "priority", "block_both"])
def test_configs4_geometry_soak_with_c_oracle(collision):
    """configs[4] geometry (256x256 maps, 256 agents = 4 waves per environment with the collision closure and the
    env-wide reductions going through LDS, obs_radius 7) over 64 steps and several episodes: 256 environments, finish
    mode with max_episode_steps 24 for all systems plus a lifelong run for 'soft'; observations every 16th step."""
    from oracle.c_oracle import COracle
    from pogema_amd import GridConfig, VecPogema
    B, size, A, r, T, max_steps = 256, 256, 256, 7, 64, 24
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, 21)
    threads = min(32, len(__import__("os").sched_getaffinity(0)))
    for on_target in (("finish", "restart") if collision == "soft" else ("finish",)):
        rng = np.random.default_rng(17)
        ref = COracle(B, size, size, A, r, collision, on_target, max_steps, True, seed=3, env_index_base=11)
        ref.reset(obstacles, agents, targets)
        gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system=collision, on_target=on_target,
                        max_episode_steps=max_steps, seed=3, density=0.3)
        env = VecPogema(gc, batch=B, auto_reset=True, env_index_base=11)
        env.reset_from_state(obstacles, agents, targets, validate=False)
        state = ref.get_state()
        for t in range(T):
            acts = goal_seeking_actions(rng, state["agents_xy"], state["targets_xy"], p_random=0.3)
            check_obs = t % 16 == 15 or t == max_steps - 1
            rout = ref.step(acts, nthreads=threads, compute_obs=check_obs)
            out = env.step(torch.from_numpy(acts).cuda(), compute_obs=check_obs)
            state = _assert_same_step(f"configs4-soak/{on_target}/{collision}", t, env, ref, out, rout, check_obs)
        env.close()
        ref.close()


INVARIANT_GEOMS = {"configs2": (8192, 64, 64, 5, 40), "configs4": (4096, 256, 256, 7, 8)}
INVARIANT_CASES = [("configs2", c, o) for c in ("priority", "block_both", "soft") for o in ("finish", "restart", "nothing")] + \
                  [("configs4", c, o) for c, o in (("soft", "finish"), ("priority", "restart"), ("block_both", "nothing"),
                                                   ("soft", "restart"))]


@pytest.mark.parametrize("geom,collision,on_target", INVARIANT_CASES, ids=["-".join(c) for c in INVARIANT_CASES])
def test_invariants_full_size(geom, collision, on_target):
    B, size, A, r, T = INVARIANT_GEOMS[geom]
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, 11)
    env, obs = _engine(B, size, A, r, collision, on_target, 16, True, obstacles, agents, targets, seed=5)
    d_obst = torch.from_numpy(obstacles).cuda()
    bi = torch.arange(B, device="cuda")[:, None]
    gen = torch.Generator(device="cuda").manual_seed(1)
    W = 2 * r + 1
    for t in range(T):
        acts = torch.randint(0, 5, (B, A), generator=gen, device="cuda")
        obs, rew, term, trunc, info = env.step(acts)
        st = env.get_state(occupancy=True)
        xy, active = st["agents_xy"].long(), st["is_active"]
        # nobody stands on an obstacle or outside the map
        assert (xy >= 0).all() and (xy < size).all()
        assert (d_obst[bi, xy[..., 0], xy[..., 1]] == 0).all()
        # visible agents occupy pairwise distinct cells; occupancy array == scatter of visible agents
        key = xy[..., 0] * size + xy[..., 1]
        key = torch.where(active, key, -1 - torch.arange(A, device="cuda")[None, :])
        srt = key.sort(dim=1).values
        assert (srt[:, 1:] != srt[:, :-1]).all()
        # (`soft`, default semantics = the literal index-order move_without_checks loop, docs/SPEC.md Q2: an agent that
        # followed a HIGHER-index agent stands on its cell but is missing from the array until a later step re-sets it)
        unseen = int(active.sum()) - int(st["occupancy"].sum())
        assert unseen >= 0 and (collision == "soft" or unseen == 0)
        # observation identities: target plane has exactly one 1; a visible agent sees itself at the centre -- the
        # unseen ones excepted -- (a hidden, finished agent may see another agent passing over its cell)
        assert torch.equal(obs[:, :, 2].sum(dim=(2, 3)), torch.ones(B, A, device="cuda"))
        assert int((obs[:, :, 1, r, r][active] == 0).sum()) == unseen
        assert (obs[:, :, 0, r, r] == 0).all()  # never inside an obstacle
        assert ((obs == 0) | (obs == 1)).all()
        # rewards are 0/1, truncation is all-or-none per env
        assert ((rew == 0) | (rew == 1)).all()
        assert (trunc.all(dim=1) | (~trunc).all(dim=1)).all()
        if on_target == "restart":
            assert not term.any()
        if on_target == "nothing":
            assert (term.all(dim=1) | (~term).all(dim=1)).all()
        el = st["elapsed"]
        assert (el >= 0).all() and (el < 16).all()  # auto-reset at the time limit
    env.close()


def test_determinism_and_reset_restores_initial_state():
    B, size, A, r = 2048, 32, 16, 5
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, 2)
    outs = []
    for rep in range(2):
        env, obs0 = _engine(B, size, A, r, "soft", "finish", 8, True, obstacles, agents, targets)
        gen = torch.Generator(device="cuda").manual_seed(4)
        trace = [obs0.clone()]
        for t in range(8):  # the 8th step truncates every env -> all reset to the initial state
            obs, *_ = env.step(torch.randint(0, 5, (B, A), generator=gen, device="cuda"))
            trace.append(obs.clone())
        st = env.get_state()
        assert np.array_equal(st["agents_xy"].cpu().numpy(), agents)
        assert np.array_equal(st["targets_xy"].cpu().numpy(), targets)
        assert st["is_active"].all() and (st["elapsed"] == 0).all()
        assert torch.equal(trace[-1], trace[0]), "observation after auto-reset == first observation"
        outs.append(torch.stack(trace))
        env.close()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("collision", ["priority", "block_both", "soft"])
@pytest.mark.parametrize("on_target", ["finish", "restart"])
def test_soak_parity_with_c_oracle(collision, on_target):
    """Long rollout (150 steps, 1024 envs x 64 agents on 64x64, several episodes per env): state, flags, rewards and
    metrics every step, full observations every 25th -- rare events (long chains, multi-way conflicts, late
    arrivals, lifelong re-targets) against the literal C port."""
    from oracle.c_oracle import COracle
    B, size, A, r, T, max_steps = 1024, 64, 64, 5, 150, 32
    obstacles, agents, targets = generate_instances(B, size, size, A, 0.3, 99)
    rng = np.random.default_rng(8)
    ref = COracle(B, size, size, A, r, collision, on_target, max_steps, True, seed=5, env_index_base=40)
    ref.reset(obstacles, agents, targets)
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system=collision, on_target=on_target,
                    max_episode_steps=max_steps, seed=5, density=0.3)
    env = VecPogema(gc, batch=B, auto_reset=True, env_index_base=40)
    env.reset_from_state(obstacles, agents, targets, validate=False)
    threads = min(32, len(__import__("os").sched_getaffinity(0)))
    for t in range(T):
        # biased walk towards the targets half of the time, so that agents actually arrive and conflict at goals
        acts = rng.integers(0, 5, size=(B, A)).astype(np.int64)
        check_obs = t % 25 == 24
        robs, rrew, rterm, rtrunc, ract = ref.step(acts, nthreads=threads, compute_obs=check_obs)
        obs, rew, term, trunc, info = env.step(torch.from_numpy(acts).cuda(), compute_obs=check_obs)
        st, rst = env.get_state(), ref.get_state()
        assert np.array_equal(st["agents_xy"].cpu().numpy(), rst["agents_xy"]), f"step {t}: positions"
        assert np.array_equal(st["targets_xy"].cpu().numpy(), rst["targets_xy"]), f"step {t}: targets"
        assert np.array_equal(st["is_active"].cpu().numpy(), rst["is_active"]) and np.array_equal(st["elapsed"].cpu().numpy(), rst["elapsed"])
        assert np.array_equal(term.cpu().numpy(), rterm) and np.array_equal(trunc.cpu().numpy(), rtrunc)
        np.testing.assert_allclose(rew.cpu().numpy(), rrew, rtol=0, atol=1e-6)
        done = info["episode_done"].cpu().numpy().astype(bool)
        assert np.array_equal(done, ref.episode_done.astype(bool))
        np.testing.assert_allclose(info["metrics"].cpu().numpy()[done], ref.metrics[done], rtol=1e-6, atol=1e-6)
        if check_obs:
            assert np.array_equal(obs.cpu().numpy(), robs), f"step {t}: observations"
    env.close()
    ref.close()
