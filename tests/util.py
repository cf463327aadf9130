"""Shared helpers for the parity tests: run one seeded scenario through the CPU oracle (test
infrastructure, oracle/) and through the HIP engine (pogema_amd, via the C-ABI) and compare."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.pogema_oracle import PogemaOracle  # noqa: E402


def generate_instances(batch, height, width, num_agents, density, seed, env_index_base=0):
    """Instances from the engine's HOST generator (pure host code; needs no GPU): env b is instance
    (seed, env_index_base + b)."""
    from pogema_amd import _lib
    lib = _lib.load()
    obstacles = np.empty((batch, height, width), dtype=np.uint8)
    agents = np.empty((batch, num_agents, 2), dtype=np.int32)
    targets = np.empty((batch, num_agents, 2), dtype=np.int32)
    _lib.check(lib.pgx_generate(batch, height, width, num_agents, float(density), int(seed), int(env_index_base), 50, 0,
                                obstacles.ctypes.data, agents.ctypes.data, targets.ctypes.data))
    return obstacles, agents, targets


def random_actions(steps, batch, num_agents, seed, p_noop=0.2):
    rng = np.random.default_rng(seed)
    a = rng.integers(1, 5, size=(steps, batch, num_agents))
    a[rng.random(a.shape) < p_noop] = 0
    return a.astype(np.int64)


def outside_density_of(obstacles):
    """`GridConfig.density` as the engine sees it for an explicit map: the obstacle fraction of env 0's map
    (what `empty_outside=False` uses beyond the border ring)."""
    m = np.asarray(obstacles[0])
    return float(sum(1 for v in m.reshape(-1) if v != 0) / m.size)


def _oracle_semantics(semantics):
    """pogema_amd.Semantics -> the oracles' keyword arguments.  None = the PRODUCT's process-wide default
    (`Semantics.from_env()`: recalled literals < pinned file < PGX_SEMANTICS): the checker runs under the semantics the
    engine it checks runs under, also after reference fixtures have pinned another default."""
    if semantics is None:
        from pogema_amd.semantics import Semantics
        semantics = Semantics.from_env()
    return semantics.oracle_kwargs()


def oracle_rollout(obstacles, agents, targets, actions, *, obs_radius, collision_system, on_target,
                   max_episode_steps, auto_reset, seed=0, env_index_base=0, empty_outside=True, semantics=None,
                   inject_targets=None):
    """Returns dict of arrays [T, B, ...] from the pure-Python oracle.  `inject_targets` [T, B, A, 2]: after step t the
    targets are overwritten with inject_targets[t] (replay of a recorded lifelong target sequence); the returned
    `targets_xy` are the ones the implementation itself produced, before the overwrite."""
    T, B, A = actions.shape
    W = 2 * obs_radius + 1
    envs = [PogemaOracle(obstacles[b], agents[b], targets[b], obs_radius=obs_radius,
                         collision_system=collision_system, on_target=on_target,
                         max_episode_steps=max_episode_steps, auto_reset=auto_reset, seed=seed,
                         env_index=env_index_base + b, empty_outside=empty_outside,
                         outside_density=outside_density_of(obstacles), **_oracle_semantics(semantics))
            for b in range(B)]
    out = {
        "obs0": np.stack([np.stack(e._obs()) for e in envs]),
        "obs": np.zeros((T, B, A, 3, W, W), np.float32), "rewards": np.zeros((T, B, A), np.float32),
        "terminated": np.zeros((T, B, A), bool), "truncated": np.zeros((T, B, A), bool),
        "is_active": np.zeros((T, B, A), bool), "agents_xy": np.zeros((T, B, A, 2), np.int32),
        "targets_xy": np.zeros((T, B, A, 2), np.int32), "elapsed": np.zeros((T, B), np.int32),
        "episode_done": np.zeros((T, B), bool), "metrics": np.zeros((T, B, 6), np.float32),
        "occupancy0": np.stack([e.get_state()["occupancy"] for e in envs]),
    }
    out["occupancy"] = np.zeros((T,) + out["occupancy0"].shape, np.uint8)  # `Grid.positions` (padded) after every step
    names = ("ISR", "CSR", "ep_length", "SoC", "makespan", "avg_throughput")
    for t in range(T):
        for b, e in enumerate(envs):
            obs, rew, term, trunc, infos = e.step(actions[t, b])
            out["obs"][t, b] = np.stack(obs)
            out["rewards"][t, b] = rew
            out["terminated"][t, b] = term
            out["truncated"][t, b] = trunc
            out["is_active"][t, b] = [i["is_active"] for i in infos]
            if "metrics" in infos[0]:
                out["episode_done"][t, b] = True
                out["metrics"][t, b] = [infos[0]["metrics"][k] for k in names]
            st = e.get_state()
            out["agents_xy"][t, b] = st["agents_xy"]
            out["targets_xy"][t, b] = st["targets_xy"]
            out["elapsed"][t, b] = st["elapsed"]
            out["occupancy"][t, b] = st["occupancy"]
            if inject_targets is not None:
                e.set_targets(inject_targets[t, b])
    return out


def c_oracle_rollout(obstacles, agents, targets, actions, *, obs_radius, collision_system, on_target,
                     max_episode_steps, auto_reset, seed=0, env_index_base=0, nthreads=1, empty_outside=True,
                     semantics=None):
    """Same rollout through the plain-C oracle port (oracle/libpogema_oracle.so)."""
    from oracle.c_oracle import COracle
    T, B, A = actions.shape
    H, Wd = obstacles.shape[1:]
    W = 2 * obs_radius + 1
    env = COracle(B, H, Wd, A, obs_radius, collision_system, on_target, max_episode_steps, auto_reset, seed,
                  env_index_base, empty_outside=empty_outside, outside_density=outside_density_of(obstacles),
                  **_oracle_semantics(semantics))
    out = {
        "obs0": env.reset(obstacles, agents, targets),
        "obs": np.zeros((T, B, A, 3, W, W), np.float32), "rewards": np.zeros((T, B, A), np.float32),
        "terminated": np.zeros((T, B, A), bool), "truncated": np.zeros((T, B, A), bool),
        "is_active": np.zeros((T, B, A), bool), "agents_xy": np.zeros((T, B, A, 2), np.int32),
        "targets_xy": np.zeros((T, B, A, 2), np.int32), "elapsed": np.zeros((T, B), np.int32),
        "episode_done": np.zeros((T, B), bool), "metrics": np.zeros((T, B, 6), np.float32),
    }
    for t in range(T):
        obs, rew, term, trunc, act = env.step(actions[t], nthreads=nthreads)
        if env.bad_action_count():  # bad_action='flag': the port counts, the host raises (like the engine)
            env.close()
            raise IndexError("an active agent's action was outside 0..4")
        out["episode_done"][t] = env.episode_done.astype(bool)
        out["metrics"][t] = np.where(out["episode_done"][t][:, None], env.metrics, 0)
        st = env.get_state()
        out["obs"][t], out["rewards"][t], out["terminated"][t], out["truncated"][t] = obs, rew, term, trunc
        out["is_active"][t] = act
        out["agents_xy"][t], out["targets_xy"][t], out["elapsed"][t] = st["agents_xy"], st["targets_xy"], st["elapsed"]
    env.close()
    return out


def engine_rollout(obstacles, agents, targets, actions, *, obs_radius, collision_system, on_target,
                   max_episode_steps, auto_reset, seed=0, env_index_base=0, action_dtype="int64",
                   device="cuda:0", obs_dtype=None, empty_outside=True, semantics=None, inject_targets=None,
                   with_occupancy=False):
    """Same rollout through the HIP engine (C-ABI via pogema_amd.VecPogema); `inject_targets` as in oracle_rollout;
    `with_occupancy`: also export the occupancy array (pgx_get_state) after the reset and after every step."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    T, B, A = actions.shape
    H, Wd = obstacles.shape[1:]
    gc = GridConfig(map=obstacles[0].tolist(), num_agents=A, obs_radius=obs_radius,
                    collision_system=collision_system, on_target=on_target, max_episode_steps=max_episode_steps,
                    seed=seed, empty_outside=empty_outside)
    extra = {} if obs_dtype is None else {"obs_dtype": obs_dtype}
    if semantics is not None:
        extra["semantics"] = semantics
    env = VecPogema(gc, batch=B, device=device, auto_reset=auto_reset, env_index_base=env_index_base, **extra)
    obs0 = env.reset_from_state(obstacles, agents, targets)
    if obs_dtype is not None:
        assert obs0.dtype == obs_dtype
    W = 2 * obs_radius + 1
    out = {
        "obs0": obs0.float().cpu().numpy() if obs0.dtype != torch.uint8 else obs0.cpu().numpy(),
        "obs": np.zeros((T, B, A, 3, W, W), np.float32), "rewards": np.zeros((T, B, A), np.float32),
        "terminated": np.zeros((T, B, A), bool), "truncated": np.zeros((T, B, A), bool),
        "is_active": np.zeros((T, B, A), bool), "agents_xy": np.zeros((T, B, A, 2), np.int32),
        "targets_xy": np.zeros((T, B, A, 2), np.int32), "elapsed": np.zeros((T, B), np.int32),
        "episode_done": np.zeros((T, B), bool), "metrics": np.zeros((T, B, 6), np.float32),
    }
    if with_occupancy:
        out["occupancy0"] = env.get_state(occupancy=True)["occupancy"].cpu().numpy()
        out["occupancy"] = np.zeros((T,) + out["occupancy0"].shape, np.uint8)
    tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[action_dtype]
    d_actions = torch.from_numpy(actions).to(device).to(tdt)
    for t in range(T):
        obs, rew, term, trunc, infos = env.step(d_actions[t])
        st = env.get_state(occupancy=with_occupancy)
        if with_occupancy:
            out["occupancy"][t] = st["occupancy"].cpu().numpy()
        out["obs"][t] = obs.float().cpu().numpy()  # uint8 / bfloat16 / float16 observations widen to float32 here (0/1: exact)
        out["rewards"][t] = rew.cpu().numpy()
        out["terminated"][t] = term.cpu().numpy()
        out["truncated"][t] = trunc.cpu().numpy()
        out["is_active"][t] = infos["is_active"].cpu().numpy()
        out["episode_done"][t] = infos["episode_done"].cpu().numpy()
        out["metrics"][t] = np.where(out["episode_done"][t][:, None], infos["metrics"].cpu().numpy(), 0)
        out["agents_xy"][t] = st["agents_xy"].cpu().numpy()
        out["targets_xy"][t] = st["targets_xy"].cpu().numpy()
        out["elapsed"][t] = st["elapsed"].cpu().numpy()
        if inject_targets is not None:
            env.set_targets(inject_targets[t])
    env.close()
    return out


def engine_rollout_launch(obstacles, agents, targets, actions, *, obs_radius, collision_system, on_target,
                          max_episode_steps, auto_reset, seed=0, env_index_base=0, action_dtype="int64",
                          device="cuda:0", obs_dtype=None, empty_outside=True, semantics=None):
    """The same rollout as ONE pgx_rollout launch (VecPogema.rollout).  Per-step positions are not exported by a
    rollout: the dict carries the per-step outputs plus the FINAL agents_xy / targets_xy / elapsed in the last slice."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    T, B, A = actions.shape
    gc = GridConfig(map=obstacles[0].tolist(), num_agents=A, obs_radius=obs_radius, collision_system=collision_system,
                    on_target=on_target, max_episode_steps=max_episode_steps, seed=seed, empty_outside=empty_outside)
    extra = {} if obs_dtype is None else {"obs_dtype": obs_dtype}
    if semantics is not None:
        extra["semantics"] = semantics
    env = VecPogema(gc, batch=B, device=device, auto_reset=auto_reset, env_index_base=env_index_base, **extra)
    obs0 = env.reset_from_state(obstacles, agents, targets)
    tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[action_dtype]
    res = env.rollout(torch.from_numpy(actions).to(device).to(tdt))
    done = res["episode_done"].cpu().numpy()
    out = {"obs0": obs0.float().cpu().numpy(), "obs": res["obs"].float().cpu().numpy(),
           "rewards": res["rewards"].cpu().numpy(), "terminated": res["terminated"].cpu().numpy(),
           "truncated": res["truncated"].cpu().numpy(), "is_active": res["is_active"].cpu().numpy(), "episode_done": done,
           "metrics": np.where(done[..., None], res["metrics"].cpu().numpy(), 0)}
    st = env.get_state()
    out["final"] = {k: st[k].cpu().numpy() for k in ("agents_xy", "targets_xy", "elapsed")}
    env.close()
    return out


def assert_rollouts_equal(ref, got, what=""):
    """Bit-exact for every integer/bool/index field and for the 0.0/1.0 float planes; rewards within
    1e-6 (BASELINE.json north_star tolerance)."""
    if "metrics" in ref and "metrics" in got:  # small-integer arithmetic in float32: exact up to one rounding
        np.testing.assert_allclose(got["metrics"], ref["metrics"], rtol=1e-6, atol=1e-6, err_msg=f"{what}: metrics")
    for key in ("agents_xy", "targets_xy", "elapsed", "terminated", "truncated", "is_active", "episode_done", "occupancy0",
                "occupancy"):
        if key not in ref or key not in got:
            continue
        if not np.array_equal(ref[key], got[key]):
            bad = np.argwhere(ref[key] != got[key])[0]
            raise AssertionError(f"{what}: {key} differs first at index {tuple(bad)}: "
                                 f"oracle={ref[key][tuple(bad)]} engine={got[key][tuple(bad)]}")
    if "rewards" in ref or "rewards" in got:  # (grid-layer fixtures carry no rewards: envs.py is not part of them)
        np.testing.assert_allclose(got["rewards"], ref["rewards"], rtol=0, atol=1e-6, err_msg=f"{what}: rewards")
    if "final" in got:  # a rollout launch exports the state only once, at its end
        for key, val in got["final"].items():
            assert np.array_equal(ref[key][-1], val), f"{what}: final {key} differs"
    for key in ("obs0", "obs"):
        if not np.array_equal(ref[key], got[key]):
            bad = np.argwhere(ref[key] != got[key])[0]
            raise AssertionError(f"{what}: {key} differs first at index {tuple(bad)}: "
                                 f"oracle={ref[key][tuple(bad)]} engine={got[key][tuple(bad)]}")


def check_spec_case(case, run):
    """One hand-derived SPEC vector (tests/golden/spec_vectors.json) against `run` (oracle_rollout, c_oracle_rollout
    or engine_rollout).  `case["semantics"]` selects non-default switches; `expect["raises"]` names the exception the
    rollout must end with."""
    import pytest
    from pogema_amd.semantics import Semantics
    obstacles = np.array(case["map"], np.uint8)[None]
    agents = np.array(case["agents_xy"], np.int32)[None]
    targets = np.array(case["targets_xy"], np.int32)[None]
    actions = np.array(case["actions"], np.int64)[:, None, :]
    kw = dict(obs_radius=case["obs_radius"], collision_system=case["collision_system"], on_target=case["on_target"],
              max_episode_steps=case.get("max_episode_steps", 64), auto_reset=False,
              semantics=Semantics(**case["semantics"]) if case.get("semantics") else None)
    exp = case["expect"]
    if "raises" in exp:
        with pytest.raises({"IndexError": IndexError}[exp["raises"]]):
            run(obstacles, agents, targets, actions, **kw)
        return
    out = run(obstacles, agents, targets, actions, **kw)
    assert out["agents_xy"][:, 0].tolist() == exp["agents_xy"], case["why"]
    if "rewards" in exp:
        assert out["rewards"][:, 0].tolist() == exp["rewards"], case["why"]
    for key in ("terminated", "truncated", "is_active"):
        if key in exp:
            assert out[key][:, 0].astype(int).tolist() == exp[key], f"{key}: {case['why']}"
    if "obs0_agent0" in exp:
        assert out["obs0"][0, 0].astype(int).tolist() == exp["obs0_agent0"]
    if "agents_plane" in exp:  # [T][A][W][W]: channel 1 (`Grid.get_positions`) of every agent's observation after each step
        assert out["obs"][:, 0, :, 1].astype(int).tolist() == exp["agents_plane"], f"agents plane: {case['why']}"


def odd_cases(seed, n):
    """(description, (obstacles, agents, targets, actions), rollout kwargs) for `n` single-environment corner cases: tiny and
    one-cell-wide maps, agents starting on their goal, shared goals, goals on other agents' starts, short time limits,
    out-of-range actions, random semantics switches, odd (seed, env index) pairs, `empty_outside` either way."""
    from pogema_amd.semantics import Semantics
    rng = np.random.default_rng(seed)
    done = 0
    while done < n:
        H, W = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        A = int(rng.integers(1, max(2, min(H * W, 9))))
        obst = (rng.random((H, W)) < float(rng.choice([0.0, 0.0, 0.15, 0.3]))).astype(np.uint8)
        free = np.argwhere(obst == 0)
        if len(free) < A:
            continue
        starts = free[rng.choice(len(free), A, replace=False)]
        targets = free[rng.integers(0, len(free), A)]
        if rng.random() < 0.3:
            k = int(rng.integers(0, A))
            targets[k] = starts[k]
        if A >= 2 and rng.random() < 0.3:
            targets[1] = targets[0]
        T = int(rng.integers(1, 9))
        actions = rng.integers(0, 5, size=(T, 1, A)).astype(np.int64)
        if rng.random() < 0.2:
            actions[rng.integers(0, T), 0, rng.integers(0, A)] = int(rng.choice([5, 7, 100]))
        kw = dict(obs_radius=int(rng.integers(1, 4)), collision_system=str(rng.choice(["priority", "block_both", "soft"])),
                  on_target=str(rng.choice(["finish", "restart", "nothing"])), max_episode_steps=int(rng.choice([1, 2, 3, 64])),
                  auto_reset=bool(rng.random() < 0.5), seed=int(rng.integers(0, 1000)), env_index_base=int(rng.integers(0, 50)),
                  empty_outside=bool(rng.random() < 0.7),
                  semantics=Semantics(soft_vertex=str(rng.choice(["lowest_index", "all_stay"])),
                                      soft_occupancy=str(rng.choice(["index_order", "exact"])),
                                      coop_reward=str(rng.choice(["all_solved", "per_agent"]))))
        what = f"odd case {done}: {H}x{W}, {A} agents, {kw}, starts {starts.tolist()}, targets {targets.tolist()}"
        yield what, (obst[None], starts[None].astype(np.int32), targets[None].astype(np.int32), actions), kw
        done += 1
