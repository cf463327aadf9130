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
