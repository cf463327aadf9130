#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native POGEMA step engine.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (`VecPogema.step` -> pgx_step -> one HIP kernel launch) over one
batch of synthetic input: BASELINE.json configs[2] -- 8192 envs per GPU, 64x64 maps, 64 agents,
obs_radius 5, density 0.3, random-obstacle maps, uniform random actions already resident in HBM.
Metric: agent-steps/sec, whole job = n_gpus * batch * agents * K / max-over-ranks wall time.
The batch shards over GPUs with no data-path collective (weak scaling); torch.distributed is used
for the start barrier and the max-over-ranks clock only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (batch per GPU, size, agents, obs_radius)  -- BASELINE.json configs[1..4]
    "cfg1": (1024, 16, 8, 5),
    "cfg2": (8192, 64, 64, 5),
    "cfg3": (8192, 32, 16, 5),   # configs[3]: 65536 envs over 8 GPUs = 8192 per GPU
    "cfg4": (4096, 256, 256, 7),
}


def algorithmic_bytes_per_agent_step(size: int, agents: int, r: int, obs_bytes: int = 4) -> float:
    """12*W^2 obs + 3*ceil(P^2/8)/A bitmaps + 21 bytes of per-agent state/IO
    (3 * obs_bytes * W^2 for the observation when the non-drop-in uint8 mode is benchmarked)."""
    W, P = 2 * r + 1, size + 2 * r
    return 3.0 * obs_bytes * W * W + 3.0 * ((P * P + 7) // 8) / agents + 21.0


def cpu_baseline(size, agents, r, collision, density, max_steps, target_seconds=12.0):
    """Times the plain-C oracle port (oracle/, kind 'port') on the host cores on a bounded sample of
    the same workload.  Reported next to the GPU number; never the thing being measured."""
    import numpy as np
    from oracle.c_oracle import COracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import generate_instances
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    B = max(256, 8 * avail)
    obstacles, agents_xy, targets_xy = generate_instances(B, size, size, agents, density, 0)
    env = COracle(B, size, size, agents, r, collision, "finish", max_steps, True)
    env.reset(obstacles, agents_xy, targets_xy)
    rng = np.random.default_rng(1)
    pool = rng.integers(0, 5, size=(16, B, agents)).astype(np.int64)
    W = 2 * r + 1
    out = (np.empty((B, agents, 3, W, W), np.float32), np.empty((B, agents), np.float32),
           np.empty((B, agents), np.uint8), np.empty((B, agents), np.uint8), np.empty((B, agents), np.uint8))
    env.step(pool[0], nthreads=1, out=out)  # touch pages
    # the host is often memory-bound on the observation write: pick the best thread count quickly
    best, cores, single = 0.0, 1, 0.0
    cands = sorted({1, 2, 4, 8, 16, 32, 64, avail // 2, avail} - {0})
    for nt in [c for c in cands if c <= avail]:
        tc = time.perf_counter()
        n = 0
        while time.perf_counter() - tc < 0.4:
            env.step(pool[n % 16], nthreads=nt, out=out)
            n += 1
        rate = n / (time.perf_counter() - tc)
        if nt == 1:
            single = rate * B * agents
        if rate > best:
            best, cores = rate, nt
    steps = 0
    t0 = time.perf_counter()
    while True:
        env.step(pool[steps % 16], nthreads=cores, out=out)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or steps >= 200000:
            break
    env.close()
    return {"value": B * agents * steps / dt, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "single_core_value": single, "host_cores_available": avail,
            "sample": f"{B} envs x {agents} agents x {steps} steps of the same workload ({size}x{size}, r={r}, "
                      f"{collision}), plain-C oracle port with OpenMP over envs, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--collision", default="soft", choices=["priority", "block_both", "soft"])
    ap.add_argument("--on-target", default="finish", choices=["finish", "restart", "nothing"])
    ap.add_argument("--batch", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--density", type=float, default=0.3)
    ap.add_argument("--max-episode-steps", type=int, default=64)
    ap.add_argument("--action-dtype", default="int64", choices=["int8", "int32", "int64"])
    ap.add_argument("--obs-dtype", default="float32", choices=["float32", "uint8"],
                    help="float32 = the reference's dtype (the headline); uint8 = the engine's lighter non-drop-in mode")
    ap.add_argument("--auto-reset", default="restore", choices=["restore", "regenerate"],
                    help="restore = finished envs return to their initial state inside the step kernel (headline); "
                         "regenerate = they get a fresh random instance on the device (pgx_regenerate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip the observation write")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from pogema_amd import GridConfig, VecPogema

    batch, size, agents, r = WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
    gc = GridConfig(size=size, density=args.density, num_agents=agents, obs_radius=r, seed=0,
                    collision_system=args.collision, on_target=args.on_target,
                    max_episode_steps=args.max_episode_steps)
    env = VecPogema(gc, batch=batch, device=device, env_index_base=rank * batch,
                    auto_reset=True if args.auto_reset == "restore" else "regenerate", reuse_buffers=True,
                    obs_dtype=torch.float32 if args.obs_dtype == "float32" else torch.uint8)
    env.reset(seed=0)
    tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[args.action_dtype]
    gen = torch.Generator(device=device)
    gen.manual_seed(1 + rank)
    pool = [torch.randint(0, 5, (batch, agents), generator=gen, device=device).to(tdt) for _ in range(32)]

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        env.step(pool[i % len(pool)], compute_obs=not args.no_obs)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # same (current) stream the engine launches on
    for i in range(args.steps):
        env.step(pool[i % len(pool)], compute_obs=not args.no_obs)
    ev1.record()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # avg launch-to-launch duration on the stream
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_agent_steps = world * batch * agents * args.steps
        value = n_agent_steps / elapsed
        bpas = algorithmic_bytes_per_agent_step(size, agents, r, 4 if args.obs_dtype == "float32" else 1)
        alg_bytes = bpas * batch * agents  # per launch (one GPU)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    key = f"{args.workload}/{args.collision}" + ("" if args.obs_dtype == "float32" else "/uint8")
                    traffic = json.load(f).get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "agent-steps/sec (whole node), 64-agent 64x64 grid, batch=8192 envs"
                      if (args.workload == "cfg2" and args.obs_dtype == "float32")
                      else f"agent-steps/sec (whole node), workload {args.workload}, obs {args.obs_dtype}",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{args.workload[-1]}]: {batch} envs/GPU, {size}x{size} map, "
                                   f"{agents} agents, obs_radius {r}, density {args.density}",
                       "collision_system": args.collision, "on_target": args.on_target, "auto_reset": args.auto_reset,
                       "max_episode_steps": args.max_episode_steps, "obs_dtype": args.obs_dtype,
                       "action_dtype": args.action_dtype, "envs_per_gpu": batch, "sharding": f"batch-sharded x{world}, no collective",
                       "obs_buffers": (f"2 alternating buffers, the fastest of {len(env.placement_us)} placement-probed "
                                       f"candidates (observation pass {env.placement_us[0]:.1f} / {env.placement_us[1]:.1f} us; "
                                       f"slowest candidate {env.placement_us[-1]:.1f} us)")
                                      if getattr(env, "placement_us", None) else "2 alternating buffers"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pgx::step_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_agent_step": bpas, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(size, agents, r, args.collision, args.density, args.max_episode_steps,
                                                args.cpu_seconds)
        print(json.dumps(line), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
