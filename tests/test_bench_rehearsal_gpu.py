"""GPU: bench.py's multi-rank flow EXECUTED with real engines on a 1-GPU box ("rehearsal").

The driver's 8-GPU run is the first time `bench.py --gpus N` meets N devices; everything around the engines -- the
launcher, rank-to-device mapping, barriers, the max-over-ranks clock, the gather of per-rank kernel times, N concurrent
zone walks, the JSON contract -- must have run before.  PGX_BENCH_SHARE_DEVICE=1 maps the ranks onto the one device
(gloo instead of RCCL, which refuses two ranks per device) and labels the line REHEARSAL; PGX_BENCH_FORCE_DIST=1 makes a
single rank go through the RCCL process group (init with device_id, barrier, all_reduce, all_gather on device tensors)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
QUICK = ["--steps", "40", "--warmup", "5", "--windows", "2", "--no-cpu-baseline", "--no-extras", "--no-default-placement"]
WITH_GATHER = [a for a in QUICK if a != "--no-extras"]  # N > 1: the only secondary figure is the host-side gather


def _check_per_rank(line, world):
    """VERDICT r4 #1a: one self-explaining record per rank."""
    pr = line["roofline"]["per_rank"]
    assert [e["rank"] for e in pr] == list(range(world))
    for e in pr:
        assert e["kernel_ms"] > 0 and e["envs"] > 0 and 0 < e["frac"] < 1.0
        assert e["gpu"]["uuid"] or e["gpu"]["pci"], "the record must say WHICH GPU"
        assert isinstance(e["spread"], bool) and "walk_candidates" in e and "policy" in e
        assert e["box_store_stream_gbs"] is None or e["box_store_stream_gbs"] > 100  # (a sanity bound: ranks may share the device)
        if e["box_store_stream_gbs"]:
            assert e["frac_of_box_store_stream"] == pytest.approx(e["achieved_gbs"] / e["box_store_stream_gbs"])
    assert [e["kernel_ms"] for e in pr] == pytest.approx(line["roofline"]["kernel_ms_per_rank"])
    assert 0 <= line["roofline"]["slowest_rank"] < world


def _check_host_gather(line, world):
    hg = line["secondary"]["host_gather"]
    assert "error" not in hg, hg
    assert hg["mode"] == ("shared segment" if world > 1 else "private staging")
    assert hg["with_gather_ms_per_step"] > 0 and hg["loop_ms_per_step"] > 0 and hg["bytes_per_step_per_rank"] > 0
    # (one rank: 53-57 GB/s; eight ranks sharing ONE device, one set of copy engines and one PCIe link: 6.7 GB/s each seen)
    assert hg["obs_d2h_equal"] is True and (5 if world == 1 else 0.3) < hg["obs_d2h_gbs"] < 70


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line, from rank 0; got {len(lines)}"
    return json.loads(lines[0])


def _check_two_rank_line(line, batch, agents):
    assert line["n_gpus"] == 2 and line["data"] == "synthetic" and line["steps"] == 40
    assert line["metric"].startswith("REHEARSAL"), "a line from ranks that share a device must not look like a measurement"
    assert line["config"]["rehearsal"] is True and line["config"]["process_group"] == "gloo"
    per_rank = line["roofline"]["kernel_ms_per_rank"]
    assert len(per_rank) == 2 and all(k > 0 for k in per_rank)
    assert line["config"]["global_batch"] == 2 * batch and line["config"]["envs_per_gpu"] == batch
    assert line["value"] == pytest.approx(2 * batch * agents * 40 / (line["ms_per_step"] * 40 * 1e-3), rel=1e-6)
    # two ranks time-share one device: a step of both takes at least the two kernels back to back would, minus overlap
    assert line["ms_per_step"] >= 0.9 * max(per_rank)
    pl = line["roofline"]["placement"]
    assert set(pl) >= {"spread", "walk_candidates", "budget_gib", "method"}


def test_self_launched_two_ranks_share_the_device():
    """`python bench.py --gpus 2` (bench.py starts the ranks itself), headline workload, both engines on cuda:0."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + WITH_GATHER, capture_output=True, text=True, timeout=900,
                       env=_env(PGX_BENCH_SHARE_DEVICE="1"))
    line = _line(p)
    _check_two_rank_line(line, 8192, 64)
    _check_per_rank(line, 2)
    _check_host_gather(line, 2)


def test_torchrun_form_two_ranks_configs3_sharded():
    """The driver's launch form (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with BASELINE
    configs[3] cut by shard_bounds (strong scaling): 16384 envs over 2 ranks."""
    from bench import _free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--workload", "cfg3", "--global-batch", "16384"] + QUICK
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(PGX_BENCH_SHARE_DEVICE="1"))
    line = _line(p)
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 16384 and line["config"]["envs_per_gpu"] == 8192
    assert line["metric"].startswith("REHEARSAL") and len(line["roofline"]["kernel_ms_per_rank"]) == 2
    assert line["value"] == pytest.approx(16384 * 16 * 40 / (line["ms_per_step"] * 40 * 1e-3), rel=1e-6)


def test_without_the_switch_two_ranks_are_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 devices")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + QUICK, capture_output=True, text=True, timeout=300, env=_env())
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_single_rank_through_rccl():
    """One rank, process group forced: RCCL init with device_id, barriers, all_reduce(MAX) and all_gather on device tensors --
    the collectives of the real multi-GPU run (timing only; nothing on the data path)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1"] + QUICK, capture_output=True, text=True, timeout=900,
                       env=_env(PGX_BENCH_FORCE_DIST="1"))
    line = _line(p)
    assert line["n_gpus"] == 1 and line["config"]["process_group"] == "nccl" and line["config"]["rehearsal"] is False
    assert line["metric"].startswith("agent-steps/sec (whole node), 64-agent 64x64 grid")
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 1
    assert 0.3 < line["roofline"]["frac"] < 1.0
    _check_per_rank(line, 1)
    assert line["roofline"]["frac_of_box_store_stream"] == line["roofline"]["per_rank"][0]["frac_of_box_store_stream"]


def test_eight_ranks_configs3_at_its_stated_size():
    """BASELINE configs[3] exactly as stated -- 65536 envs sharded over EIGHT ranks (strong scaling, shard_bounds) -- with
    eight real processes and engines, all on the one device of the box: the launch the driver's 8-GPU node will see, minus
    the seven other GPUs."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--workload", "cfg3", "--global-batch", "65536"] + WITH_GATHER,
                       capture_output=True, text=True, timeout=900, env=_env(PGX_BENCH_SHARE_DEVICE="1"))
    line = _line(p)
    _check_per_rank(line, 8)
    _check_host_gather(line, 8)
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["metric"].startswith("REHEARSAL (8 ranks share")
    assert line["config"]["global_batch"] == 65536 and line["config"]["envs_per_gpu"] == 8192
    per_rank = line["roofline"]["kernel_ms_per_rank"]
    assert len(per_rank) == 8 and all(k > 0 for k in per_rank)
    assert line["value"] == pytest.approx(65536 * 16 * 40 / (line["ms_per_step"] * 40 * 1e-3), rel=1e-6)


def test_a_failed_walk_is_paid_once_by_the_whole_bench():
    """VERDICT r4 #1b at the level of bench.py: on a GPU where the walk finds nothing (forced here: PGX_ZONE_SCAN accepts no
    candidate) the headline engine walks ONCE; the secondary engines (two pipelined halves, the rollout ring) reuse the
    verdict instead of holding the memory three more times."""
    args = ["--steps", "40", "--warmup", "5", "--windows", "2", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=900,
                       env=_env(PGX_ZONE_SCAN="1", PGX_ZONE_SPACER_GIB="24"))
    line = _line(p)
    rf = line["roofline"]
    assert rf["zone_walks_in_process"] == {"walks": 1, "failed": 1}, rf["zone_walks_in_process"]
    assert rf["placement"]["spread"] is False and rf["placement"]["walk_candidates"] == 3
    assert rf["box_store_stream_gbs"] and rf["frac_of_box_store_stream"] == pytest.approx(rf["achieved"] / rf["box_store_stream_gbs"])
    sec = line["secondary"]
    assert "errors" not in sec, sec.get("errors")
    assert {"pipelined", "rollout", "host_gather"} <= set(sec)
    _check_host_gather(line, 1)
    _check_per_rank(line, 1)
