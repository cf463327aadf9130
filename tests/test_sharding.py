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


def _gather_worker(rank, world, port, tmpdir, shared):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pogema_amd.sharding import HostGather, gather_to_host, shard_bounds
        GB, A = 11, 5  # ragged: 6 + 5 rows
        start, count = shard_bounds(GB, world, rank)
        fields = {"rewards": ((A,), torch.float32), "terminated": ((A,), torch.bool), "truncated": ((A,), torch.bool),
                  "is_active": ((A,), torch.bool), "episode_done": ((), torch.bool), "metrics": ((6,), torch.float32),
                  "obs": ((A, 3, 3, 3), torch.float32)}
        g = HostGather(fields, GB, shared=shared)  # 3 slots: the pipelined loop below needs them
        assert g.slots == 3
        assert g.mode == ("shared segment" if shared else "private staging + gloo gather")
        assert not os.path.exists("/dev/shm") or not [n for n in os.listdir("/dev/shm") if n.startswith("pgx_gather_")], \
            "the segment's name must be gone once everybody has mapped it"
        gen = torch.Generator().manual_seed(5)
        history = []
        for t in range(5):
            full = {"rewards": torch.rand((GB, A), generator=gen), "terminated": torch.rand((GB, A), generator=gen) > 0.5,
                    "truncated": torch.rand((GB, A), generator=gen) > 0.5, "is_active": torch.rand((GB, A), generator=gen) > 0.5,
                    "episode_done": torch.rand((GB,), generator=gen) > 0.5, "metrics": torch.rand((GB, 6), generator=gen),
                    "obs": torch.rand((GB, A, 3, 3, 3), generator=gen)}
            # the engine's recycled outputs: rewards | terminated | truncated | is_active carved from ONE block
            n = count * A
            block = torch.empty(7 * n + 16, dtype=torch.uint8)
            mine = {"rewards": block[:4 * n].view(torch.float32).view(count, A),
                    "terminated": block[4 * n:5 * n].view(torch.bool).view(count, A),
                    "truncated": block[5 * n:6 * n].view(torch.bool).view(count, A),
                    "is_active": block[6 * n:7 * n].view(torch.bool).view(count, A)}
            for k in mine:
                mine[k].copy_(full[k][start:start + count])
            for k in ("episode_done", "metrics", "obs"):
                mine[k] = full[k][start:start + count].contiguous()
            history.append(full)
            ticket = g.start(**mine)
            # one rank: the four carved outputs are one copy; several ranks: rows of other ranks lie in between
            assert g.copies_per_step == 7
            if t >= 1:  # pipelined: finish the previous step while this one is "in flight"
                got = g.finish(ticket - 1)
                if rank == 0:
                    for k, want in history[t - 1].items():
                        assert got[k].dtype == want.dtype and torch.equal(got[k], want), f"step {t - 1} field {k}"
                else:
                    assert got is None
        got = g.finish()
        if rank == 0:
            for k, want in history[-1].items():
                assert torch.equal(got[k], want)
        # the same bytes as the plain gather_to_host
        ref = gather_to_host(history[-1]["rewards"][start:start + count], GB)
        if rank == 0:
            assert torch.equal(ref, got["rewards"])
            with pytest.raises(ValueError):
                g.finish(0)  # long overwritten
            open(os.path.join(tmpdir, f"ok{int(shared)}"), "w").write("ok")
        g.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shared", [True, False])
def test_host_gather_two_ranks(tmp_path, shared):
    """HostGather over gloo, world 2, ragged slices: every rank's rows land in place in rank 0's global tensors -- through
    the shared segment (no collective on the data path) and through the private-staging + gloo fallback; pipelined
    finish(ticket - 1); the segment leaves no name behind in /dev/shm."""
    port = 29500 + ((os.getpid() + 7 + int(shared)) % 2000)
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    assert (tmp_path / f"ok{int(shared)}").exists()


def test_host_gather_single_process_merges_adjacent_fields():
    """One rank (no process group): fields that lie back to back in the source storage and in the slot travel as ONE copy
    (the recycled output block: 7 bytes per agent), stragglers on their own; slots rotate; a stale ticket is refused."""
    from pogema_amd.sharding import HostGather
    B, A = 6, 4
    n = B * A
    fields = {"rewards": ((A,), torch.float32), "terminated": ((A,), torch.bool), "truncated": ((A,), torch.bool),
              "is_active": ((A,), torch.bool), "episode_done": ((), torch.bool), "metrics": ((6,), torch.float32)}
    g = HostGather(fields, B, slots=2)
    assert g.mode == "private staging" and g.world == 1
    outs = []
    for t in range(3):
        block = torch.randint(0, 2, (7 * n + 16,), dtype=torch.uint8)
        mine = {"rewards": block[:4 * n].view(torch.float32).view(B, A), "terminated": block[4 * n:5 * n].view(torch.bool).view(B, A),
                "truncated": block[5 * n:6 * n].view(torch.bool).view(B, A), "is_active": block[6 * n:7 * n].view(torch.bool).view(B, A),
                "episode_done": torch.rand(B) > 0.5, "metrics": torch.rand(B, 6)}
        tk = g.start(**mine)
        assert g.copies_per_step == 3  # block, episode_done, metrics
        got = g.finish(tk)
        for k in fields:
            assert torch.equal(got[k], mine[k])
        outs.append((tk, {k: v.clone() for k, v in mine.items()}, got))
    # slot of step 0 was rewritten by step 2; step 1's views are still intact
    assert torch.equal(outs[1][2]["metrics"], outs[1][1]["metrics"])
    assert torch.equal(outs[0][2]["metrics"], outs[2][1]["metrics"])
    with pytest.raises(ValueError):
        g.finish(0)
    # two slots carry an unpipelined loop only: a second start() without a finish() in between would overwrite a slot
    # rank dst may still be reading
    g.start(**mine)
    with pytest.raises(ValueError, match="views are still valid"):
        g.start(**mine)
    g.finish()
    with pytest.raises(ValueError):
        g.start(rewards=mine["rewards"])
    with pytest.raises(ValueError):
        g.start(**dict(mine, metrics=torch.rand(B, 5)))
    g.close()


def _ring_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        import time
        from pogema_amd.sharding import HostGather, shard_bounds
        GB, A, T = 23, 3, 120
        start, count = shard_bounds(GB, world, rank)
        g = HostGather({"x": ((A,), torch.float32), "f": ((), torch.bool)}, GB, slots=3)
        assert g.mode == "shared segment"
        rnd = random.Random(rank)
        rows = torch.arange(start, start + count, dtype=torch.float32)[:, None] * 1000.0

        def payload(t):
            return {"x": (rows + t + torch.arange(A, dtype=torch.float32)[None]).contiguous(),
                    "f": ((torch.arange(start, start + count) + t) % 3 == 0)}

        prev = None
        for t in range(T):
            if rank == 0 and t % 7 == 3:
                time.sleep(0.004)   # a slow reader: the writers must wait for its release, not overwrite
            elif rank != 0 and rnd.random() < 0.2:
                time.sleep(0.002)   # a slow writer: the reader must wait for its rows, not read stale ones
            tk = g.start(**payload(t))
            if prev is not None:
                got = g.finish(prev)
                if rank == 0:
                    want_x = torch.arange(GB, dtype=torch.float32)[:, None] * 1000.0 + prev + torch.arange(A, dtype=torch.float32)[None]
                    want_f = (torch.arange(GB) + prev) % 3 == 0
                    if t % 5 == 0:
                        time.sleep(0.001)  # ... and the views must still be intact after a pause
                    assert torch.equal(got["x"], want_x), f"ticket {prev}: rows of another step"
                    assert torch.equal(got["f"], want_f)
            prev = tk
        g.finish(prev)
        dist.barrier()
        g.close()
        if rank == 0:
            open(os.path.join(tmpdir, "ring_ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_host_gather_ring_protocol_three_ranks(tmp_path):
    """The shared-segment ring under uneven speeds: three ranks, 120 pipelined steps, a reader that stalls and writers that
    stall -- every ticket's global tensors must hold exactly that step's rows (landed / released counters, no gloo call on
    the per-step path)."""
    port = 29500 + ((os.getpid() + 31) % 2000)
    mp.spawn(_ring_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert (tmp_path / "ring_ok").exists()


def _one_rank_fails_worker(rank, world, port, tmpdir, where):
    """the shared segment cannot be set up on ONE rank only -- every rank must still run the same collectives
    and end up on the private-staging fallback together, quickly (no process-group timeout)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        import time
        import pogema_amd.sharding as sh
        if rank == 1:
            if where == "mmap":
                import mmap as _mmap

                def boom(*a, **k):
                    raise OSError("mmap refused on this rank (test)")
                _mmap_mmap, _mmap.mmap = _mmap.mmap, boom
            else:  # the failure sits between the two barriers: counter initialisation / hipHostRegister
                real = torch.frombuffer

                def boom(*a, **k):
                    raise RuntimeError("registration refused on this rank (test)")
                torch.frombuffer = boom
        t0 = time.monotonic()
        g = sh.HostGather({"rewards": ((3,), torch.float32)}, 7, shared=True)
        took = time.monotonic() - t0
        if rank == 1:
            if where == "mmap":
                _mmap.mmap = _mmap_mmap
            else:
                torch.frombuffer = real
        assert g.mode == "private staging + gloo gather", g.mode
        assert took < 30.0, f"fallback took {took:.1f} s: the ranks waited for a timeout"
        start, count = sh.shard_bounds(7, world, rank)
        full = torch.arange(21, dtype=torch.float32).view(7, 3)
        got = g.finish(g.start(rewards=full[start:start + count].contiguous()))
        if rank == 0:
            assert torch.equal(got["rewards"], full)
            open(os.path.join(tmpdir, "ok_" + where), "w").write("ok")
        g.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("where", ["mmap", "register"])
def test_host_gather_falls_back_together_when_one_rank_fails(tmp_path, where):
    port = 29500 + ((os.getpid() + 977 + (13 if where == "mmap" else 0)) % 2000)
    mp.spawn(_one_rank_fails_worker, args=(2, port, str(tmp_path), where), nprocs=2, join=True)
    assert (tmp_path / ("ok_" + where)).exists()
