"""CPU: bench.py's launch contract.  `--gpus N` without torchrun must start N ranks itself or
fail loudly -- never print a line for fewer GPUs than asked.  The multi-rank flow (barriers, max-over-ranks clock,
per-rank kernel times, median of windows, the JSON contract) is driven with `--stub` (a host sleep instead of the
engine, gloo instead of RCCL; such a line says data = "stub")."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_n_without_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 devices")
    p = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert p.returncode != 0
    assert "HIP device" in p.stderr
    assert p.stdout.strip() == "", "no JSON line may be printed for a run that did not use the requested GPUs"


def test_world_size_must_match_gpus():
    p = _run(["--gpus", "2", "--stub"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "must agree" in p.stderr + p.stdout


def test_self_launched_two_ranks_stub():
    p = _run(["--gpus", "2", "--stub", "--steps", "20", "--warmup", "2", "--windows", "3"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 2 and line["data"] == "stub"
    assert line["metric"].startswith("STUB")
    assert line["windows"]["n"] == 3 and len(line["windows"]["ms_per_step"]) == 3
    # rank r sleeps 0.5 ms * (r + 1) per step: the clock is the MAX over ranks, i.e. rank 1's ~1 ms
    assert line["ms_per_step"] >= 0.95
    assert sorted(line["windows"]["ms_per_step"])[1] == pytest.approx(line["ms_per_step"])
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 2
    assert line["roofline"]["kernel_ms_per_rank"][1] > line["roofline"]["kernel_ms_per_rank"][0]
    # weak scaling: 8192 envs per rank, value counts both ranks
    assert line["scaling"] == "weak" and line["config"]["global_batch"] == 2 * 8192
    assert line["value"] == pytest.approx(2 * 8192 * 64 * 20 / (line["ms_per_step"] * 20 * 1e-3), rel=1e-6)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line
    assert "model" not in line["config"] and "workload" in line["config"]


def test_global_batch_is_sharded_strong_scaling():
    p = _run(["--gpus", "2", "--stub", "--steps", "4", "--warmup", "0", "--windows", "1", "--workload", "cfg3",
              "--global-batch", "11"])
    assert p.returncode == 0, p.stderr
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 11 and line["config"]["envs_per_gpu"] == 6
    assert line["value"] == pytest.approx(11 * 16 * 4 / (line["ms_per_step"] * 4 * 1e-3), rel=1e-6)


def test_under_torchrun_env_single_rank_stub():
    """The driver's form: the process IS a rank (WORLD_SIZE etc. in the environment); nothing is spawned."""
    from bench import _free_port
    p = _run(["--gpus", "1", "--stub", "--steps", "3", "--warmup", "0", "--windows", "1"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(_free_port())})
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def _stub_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_per_rank_records_are_gathered():
    """the line carries one record per rank (rank, envs, kernel time, host; on a GPU also which GPU, where its
    buffers lie and what a bare store stream does there) and names the slowest rank -- the one that sets the node's clock."""
    line = _stub_line(_run(["--gpus", "3", "--stub", "--steps", "10", "--warmup", "1", "--windows", "2", "--workload", "cfg3",
                            "--global-batch", "10"]))
    pr = line["roofline"]["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1, 2] and [e["envs"] for e in pr] == [4, 3, 3]
    assert all(e["kernel_ms"] > 0 and e["host"] for e in pr)
    assert [e["kernel_ms"] for e in pr] == pytest.approx(line["roofline"]["kernel_ms_per_rank"])
    assert line["roofline"]["slowest_rank"] == 2  # rank r sleeps 0.5 ms * (r + 1) per step
    assert line["config"]["process_group"] == "gloo"


@pytest.mark.parametrize("fake", ["", "1"])
def test_rccl_fallback_is_a_collective_decision(fake):
    """whether RCCL or gloo carries the clock is decided by ALL ranks together (all_reduce MIN over the gloo
    default group), never per rank inside an except block.  No GPU here, so RCCL fails -- on every rank, or (faked) on
    rank 1 before rank 0 has even tried: both ways every rank ends up on gloo, the line says why, nobody hangs."""
    p = _run(["--gpus", "2", "--stub", "--stub-rccl", "--steps", "5", "--warmup", "0", "--windows", "1"],
             env_extra={"PGX_BENCH_FAKE_RCCL_FAIL": fake, "PGX_BENCH_GROUP_TIMEOUT": "120"}, timeout=280)
    line = _stub_line(p)
    label = line["config"]["process_group"]
    assert label.startswith("gloo (RCCL unusable on ") and "of 2 ranks" in label
    assert "RCCL is not usable on rank" in p.stderr
    assert line["n_gpus"] == 2 and len(line["roofline"]["per_rank"]) == 2


def test_default_form_carries_every_baseline_config():
    """the driver's one line (N = 1, default workload) carries configs[1], the configs[3] shard and
    configs[4] under secondary.workloads with a fixed key set, and the configs[0] shape (size 8, 2 agents, r 3, ONE env) on
    the CPU under cpu_baseline.configs0_python_literal.  Stub mode pins the keys; the values are measured on the GPU box."""
    from bench import WORKLOAD_KEYS
    line = _stub_line(_run(["--gpus", "1", "--stub", "--steps", "5", "--warmup", "1", "--windows", "1"]))
    wl = line["secondary"]["workloads"]
    assert sorted(wl) == ["cfg1", "cfg3", "cfg4"]
    for name, rec in wl.items():
        for key in WORKLOAD_KEYS:
            assert key in rec, (name, key)
        assert ("graph_ms_per_step" in rec) == (name in ("cfg1", "cfg3")), name
    c0 = line["cpu_baseline"]["configs0_python_literal"]
    assert c0["value"] > 0 and c0["cores"] == 1 and "size=8, num_agents=2, obs_radius=3" in c0["sample"]
    assert c0["c_port"]["value"] > 0
    # a non-default workload does not drag the others along
    other = _stub_line(_run(["--gpus", "1", "--stub", "--steps", "5", "--warmup", "1", "--windows", "1", "--workload", "cfg3"]))
    assert "workloads" not in (other.get("secondary") or {})


def test_launcher_counts_devices_from_sysfs_not_hip(monkeypatch, tmp_path):
    """the self-launcher must provably not initialise HIP before it spawns its ranks: devices are counted
    from the KFD topology in sysfs (narrowed by *_VISIBLE_DEVICES); torch is only asked where there is no KFD at all."""
    import glob as _glob
    import bench
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024, 1024)):  # node 0 = the CPU
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\n")
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat, *a, **k: sorted(str(p) for p in nodes.iterdir())
                        if pat.startswith("/sys/class/kfd/kfd/topology/nodes/") else real(pat, *a, **k))
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("HIP must not be asked")))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_device_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_device_count() == 2
