"""GPU: the N > 1 path with REAL HIP engines on the one device a test box has.

(a) BASELINE.json configs[3] at its stated size -- 65536 environments of 32x32 / 16 agents -- as the 8 shards
    `sharding.shard_bounds(65536, 8, r)` gives the 8 GPUs of a node, run one after the other on cuda:0: every shard
    engine == its slice of the plain-C oracle == its slice of ONE 65536-env engine, in lifelong (`restart`) mode so that
    the per-agent target streams (keyed by the GLOBAL env index, `env_index_base`) matter.
(b) two processes over gloo, both on cuda:0: each rank builds `make_sharded_env`, steps its slice, and
    `gather_to_host` assembles real engine outputs on rank 0, which compares them with the unsharded engine and the oracle.
Environments never interact, so there is no data-path collective anywhere: what these tests pin is that a shard's
results do not depend on how the batch was cut."""
import os
import sys

import numpy as np
import pytest
import torch

from util import generate_instances

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _goal_seeking(rng, agents_xy, targets_xy, p_random=0.4):
    d = targets_xy.astype(np.int64) - agents_xy.astype(np.int64)
    along_x = np.abs(d[..., 0]) >= np.abs(d[..., 1])
    greedy = np.where(along_x, np.where(d[..., 0] < 0, 1, 2), np.where(d[..., 1] < 0, 3, 4))
    greedy = np.where((d == 0).all(axis=-1), 0, greedy)
    rnd = rng.integers(0, 5, size=greedy.shape)
    return np.where(rng.random(greedy.shape) < p_random, rnd, greedy).astype(np.int8)


def test_configs3_full_batch_as_eight_shards():
    from oracle.c_oracle import COracle
    from pogema_amd import GridConfig, VecPogema
    from pogema_amd.sharding import shard_bounds
    GB, WORLD, size, A, r, T, max_steps = 65536, 8, 32, 16, 5, 10, 6
    obstacles, agents, targets = generate_instances(GB, size, size, A, 0.3, 31)
    gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system="soft", on_target="restart",
                    max_episode_steps=max_steps, seed=13, density=0.3)
    ref = COracle(GB, size, size, A, r, "soft", "restart", max_steps, True, seed=13, env_index_base=0)
    ref_obs = ref.reset(obstacles, agents, targets)
    whole = VecPogema(gc, batch=GB, auto_reset=True, env_index_base=0)
    obs = whole.reset_from_state(obstacles, agents, targets, validate=False)
    assert np.array_equal(obs.cpu().numpy(), ref_obs)
    spans = [shard_bounds(GB, WORLD, k) for k in range(WORLD)]
    assert spans == [(8192 * k, 8192) for k in range(WORLD)]
    shards = []
    for start, count in spans:
        env = VecPogema(gc, batch=count, auto_reset=True, env_index_base=start)
        first = env.reset_from_state(obstacles[start:start + count], agents[start:start + count],
                                     targets[start:start + count], validate=False)
        assert torch.equal(first, obs[start:start + count])
        shards.append(env)
    shape = lambda e: {k: v for k, v in e.geometry().items() if k != "grid"}  # noqa: E731  (grid: per-engine XCD share tuning)
    assert shape(shards[0]) == shape(shards[-1]), "every rank runs the same launch shape"
    del obs, ref_obs, first
    rng = np.random.default_rng(5)
    threads = min(32, len(os.sched_getaffinity(0)))
    state = ref.get_state()
    retargets = 0
    obs_steps = {0, max_steps - 1, T - 1}  # first step, the auto-reset step, the last step
    for t in range(T):
        acts = _goal_seeking(rng, state["agents_xy"], state["targets_xy"])
        check_obs = t in obs_steps
        robs, rrew, rterm, rtrunc, ract = ref.step(acts, nthreads=threads, compute_obs=check_obs)
        d_acts = torch.from_numpy(acts).cuda()
        wobs, wrew, wterm, wtrunc, winfo = whole.step(d_acts, compute_obs=check_obs)
        state = ref.get_state()
        wst = whole.get_state()
        assert np.array_equal(wst["agents_xy"].cpu().numpy(), state["agents_xy"]), f"step {t}: whole-batch positions"
        assert np.array_equal(wst["targets_xy"].cpu().numpy(), state["targets_xy"]), f"step {t}: whole-batch targets"
        np.testing.assert_allclose(wrew.cpu().numpy(), rrew, rtol=0, atol=1e-6)
        if check_obs:
            assert np.array_equal(wobs.cpu().numpy(), robs), f"step {t}: whole-batch observations"
        retargets += int(rrew.sum())
        for k, (start, count) in enumerate(spans):
            sl = slice(start, start + count)
            sobs, srew, sterm, strunc, sinfo = shards[k].step(d_acts[sl], compute_obs=check_obs)
            sst = shards[k].get_state()
            what = f"step {t}, shard {k} (envs {start}..{start + count - 1})"
            assert torch.equal(sst["agents_xy"], wst["agents_xy"][sl]), f"{what}: positions"
            assert torch.equal(sst["targets_xy"], wst["targets_xy"][sl]), f"{what}: lifelong targets depend on the sharding"
            assert torch.equal(sst["elapsed"], wst["elapsed"][sl]) and torch.equal(sst["is_active"], wst["is_active"][sl])
            assert torch.equal(srew, wrew[sl]) and torch.equal(sterm, wterm[sl]) and torch.equal(strunc, wtrunc[sl]), what
            assert torch.equal(sinfo["episode_done"], winfo["episode_done"][sl]), what
            done = sinfo["episode_done"]
            assert torch.equal(sinfo["metrics"][done], winfo["metrics"][sl][done]), f"{what}: metrics"
            if check_obs:
                assert torch.equal(sobs, wobs[sl]), f"{what}: observations"
            del sobs
        del wobs, robs
    assert retargets > 1000, "the rollout must contain lifelong re-targets (otherwise the streams were not exercised)"
    for env in shards:
        env.close()
    whole.close()
    ref.close()


def _rank(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pogema_amd import GridConfig, VecPogema
        from pogema_amd.sharding import (HostGather, gather_to_host, make_sharded_env, shard_bounds, start_step_gather,
                                         step_output_fields)
        from util import generate_instances, random_actions
        torch.cuda.set_device(0)  # both ranks share the one device of the box
        GB, size, A, r, T = 1001, 24, 12, 4, 14  # a global batch the world size does not divide
        gc = GridConfig(size=size, num_agents=A, obs_radius=r, collision_system="soft", on_target="restart",
                        max_episode_steps=5, seed=21, density=0.25)
        start, count = shard_bounds(GB, world, rank)
        env = make_sharded_env(gc, GB, auto_reset=True)
        assert (env.batch, env.env_index_base) == (count, start)
        obs0, _ = env.reset(seed=77)  # on-device generation: env i draws instance (77, global index i)
        actions = torch.from_numpy(random_actions(T, GB, A, 3)).to(torch.int8)
        mine = actions[:, start:start + count].cuda()
        rew_sum = torch.zeros((count, A), device="cuda")
        # the engineered gather rides on the loop: every step's outputs (observation included here) DMA'd into the shared
        # page-locked segment by both ranks, finished one step late (pipelined) -- and the plain gather_to_host of the same
        # tensors must return the same bytes
        hg = HostGather(step_output_fields(env, with_obs=True), GB)  # (3 slots: start(t), then finish(t - 1))
        assert hg.mode == "shared segment" and hg._host.is_pinned()
        prev, per_step = None, []
        for t in range(T):
            res = env.step(mine[t])
            obs, rew, term, trunc, info = res
            rew_sum += rew
            ticket = start_step_gather(hg, res, with_obs=True)
            plain = {"rewards": gather_to_host(rew, GB), "terminated": gather_to_host(term, GB),
                     "truncated": gather_to_host(trunc, GB), "is_active": gather_to_host(info["is_active"], GB),
                     "episode_done": gather_to_host(info["episode_done"], GB), "metrics": gather_to_host(info["metrics"], GB),
                     "obs": gather_to_host(obs, GB)}
            per_step.append(plain)
            if prev is not None:
                got_prev = hg.finish(prev)
                if rank == 0:
                    for k, want in per_step[prev].items():
                        assert got_prev[k].dtype == want.dtype and torch.equal(got_prev[k], want), f"HostGather step {prev}: {k}"
                else:
                    assert got_prev is None
                per_step[prev] = None
            prev = ticket
        last = hg.finish(prev)
        if rank == 0:
            for k, want in per_step[prev].items():
                assert torch.equal(last[k], want), f"HostGather last step: {k}"
        del last
        hg.close()
        st = env.get_state()
        got = {k: gather_to_host(v, GB) for k, v in (("obs0", obs0), ("obs", obs), ("rew_sum", rew_sum), ("trunc", trunc),
                                                     ("xy", st["agents_xy"]), ("tgt", st["targets_xy"]),
                                                     ("elapsed", st["elapsed"]), ("map", env._initial[0]))}
        # the same with two steps in flight, uneven host speeds and only the small outputs (the form bench.py times): every
        # ticket's host tensors must be exactly that step's -- snapshot buffers, slots and recycled output sets are all reused
        import random
        import time
        rnd = random.Random(100 + rank)
        hg2 = HostGather(step_output_fields(env), GB, slots=4)
        pend, want = [], {}
        for t in range(T, T + 30):
            res = env.step(mine[t % T])
            tk = start_step_gather(hg2, res)
            want[tk] = {"rewards": gather_to_host(res[1], GB), "terminated": gather_to_host(res[2], GB),
                        "is_active": gather_to_host(res[4]["is_active"], GB),
                        "episode_done": gather_to_host(res[4]["episode_done"], GB), "metrics": gather_to_host(res[4]["metrics"], GB)}
            del res
            pend.append(tk)
            if rnd.random() < 0.3:
                time.sleep(0.002 * (1 + rank))
            if len(pend) > 2:
                k = pend.pop(0)
                got_k = hg2.finish(k)
                if rank == 0:
                    for name, w in want.pop(k).items():
                        assert torch.equal(got_k[name], w), f"HostGather (two steps in flight) ticket {k}: {name}"
        while pend:
            k = pend.pop(0)
            got_k = hg2.finish(k)
            if rank == 0:
                for name, w in want.pop(k).items():
                    assert torch.equal(got_k[name], w), f"HostGather (two steps in flight) ticket {k}: {name}"
        hg2.close()
        env.close()
        if rank == 0:
            # the unsharded engine on the same device ...
            ref = VecPogema(gc, batch=GB, auto_reset=True, env_index_base=0)
            robs0, _ = ref.reset(seed=77)
            d_actions = actions.cuda()
            rsum = torch.zeros((GB, A), device="cuda")
            for t in range(T):
                robs, rrew, rterm, rtrunc, rinfo = ref.step(d_actions[t])
                rsum += rrew
            rst = ref.get_state()
            for key, want in (("obs0", robs0), ("obs", robs), ("rew_sum", rsum), ("trunc", rtrunc), ("xy", rst["agents_xy"]),
                              ("tgt", rst["targets_xy"]), ("elapsed", rst["elapsed"]), ("map", ref._initial[0])):
                assert torch.equal(got[key], want.cpu()), f"gathered {key} differs from the unsharded engine"
            # ... and the CPU oracle on the gathered instances
            from oracle.c_oracle import COracle
            orc = COracle(GB, size, size, A, r, "soft", "restart", 5, True, seed=21, env_index_base=0)
            o0 = orc.reset(ref._initial[0].cpu().numpy(), ref._initial[1].cpu().numpy(), ref._initial[2].cpu().numpy())
            assert np.array_equal(o0, got["obs0"].numpy())
            for t in range(T):
                oobs, *_ = orc.step(actions[t].numpy().astype(np.int64), nthreads=4)
            assert np.array_equal(oobs, got["obs"].numpy()), "gathered observations differ from the oracle"
            assert np.array_equal(orc.get_state()["targets_xy"], got["tgt"].numpy())
            assert float(got["rew_sum"].sum()) > 0, "no lifelong arrival in the rollout"
            ref.close()
            orc.close()
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_engines_gather_to_host(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
