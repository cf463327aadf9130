"""CPU, world_size 2 over gloo: the N > 1 path's host logic -- contiguous batch slices, per-env
seeding independent of the sharding, host-side gather, and (with the C oracle standing in for the
device engine, which needs a GPU) sharded stepping == unsharded stepping."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_batch():
    from pogema_amd.sharding import shard_bounds
    for batch, world in ((65536, 8), (10, 4), (7, 8), (8192, 1), (5, 2)):
        spans = [shard_bounds(batch, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == batch
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.c_oracle import COracle
        from pogema_amd.sharding import gather_to_host, shard_bounds
        from util import generate_instances, random_actions
        GB, H, W, A, r, T = 11, 10, 10, 6, 3, 12  # global batch not divisible by the world size
        start, count = shard_bounds(GB, world, rank)
        seed = 321
        # this rank's slice: global env i draws instance (seed, i), exactly like VecPogema.generate()
        obstacles, agents, targets = generate_instances(count, H, W, A, 0.2, seed, env_index_base=start)
        actions = random_actions(T, GB, A, 5)[:, start:start + count]
        env = COracle(count, H, W, A, r, "soft", "restart", 5, True, seed=9, env_index_base=start)
        env.reset(obstacles, agents, targets)
        for t in range(T):
            obs, rew, term, trunc, act = env.step(actions[t])
        state = env.get_state()
        got_obs = gather_to_host(torch.from_numpy(obs), GB)
        got_xy = gather_to_host(torch.from_numpy(state["agents_xy"]), GB)
        got_tgt = gather_to_host(torch.from_numpy(state["targets_xy"]), GB)
        got_map = gather_to_host(torch.from_numpy(obstacles), GB)
        got_flags = gather_to_host(torch.from_numpy(np.ascontiguousarray(act)), GB)  # bool: gloo has none, travels as bytes
        if rank == 0:
            # the unsharded run
            o1, a1, t1 = generate_instances(GB, H, W, A, 0.2, seed)
            ref = COracle(GB, H, W, A, r, "soft", "restart", 5, True, seed=9, env_index_base=0)
            ref.reset(o1, a1, t1)
            full_actions = random_actions(T, GB, A, 5)
            for t in range(T):
                robs, _, _, _, ract = ref.step(full_actions[t])
            assert got_flags.dtype == torch.bool and np.array_equal(got_flags.numpy(), ract)
            rstate = ref.get_state()
            assert np.array_equal(got_map.numpy(), o1)
            assert np.array_equal(got_xy.numpy(), rstate["agents_xy"])
            assert np.array_equal(got_tgt.numpy(), rstate["targets_xy"]), "lifelong streams must not depend on sharding"
            assert np.array_equal(got_obs.numpy(), robs)
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_rollout_equals_unsharded(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
