#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native POGEMA step engine.

    python bench.py --gpus N --steps K --warmup W [--windows 5] [--workload cfg2] [--global-batch G]

A "step" is one pass of the hot path (`VecPogema.step` -> pgx_step -> one HIP kernel launch) over one batch of
synthetic input: BASELINE.json configs[2] by default -- 8192 envs per GPU, 64x64 maps, 64 agents, obs_radius 5,
density 0.3, random-obstacle maps, uniform random actions already resident in HBM.
Metric: agent-steps/sec, whole job = (envs over all ranks) * agents * K / max-over-ranks wall time of K steps.

Launch contract:
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` every process is one rank
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment); WORLD_SIZE must equal --gpus;
  * `python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts the N ranks ITSELF (one child process per device,
    before this process touches the GPU) and fails loudly when the box has fewer than N devices -- it never prints a
    line whose n_gpus differs from --gpus;
  * the batch shards over the ranks with no data-path collective; the default process group is gloo (agreement,
    per-rank records, set-up of the host gather), an RCCL group on top of it carries the barriers, the max-over-ranks
    clock and the per-rank kernel times -- or gloo does, if RCCL is unusable on ANY rank (decided collectively,
    setup_groups); every rank's GPU, placement and kernel time is in the line (roofline.per_rank).
Rehearsal: the builder's boxes have ONE GPU, so the N > 1 flow cannot be measured there -- but it
can be EXECUTED: with PGX_BENCH_SHARE_DEVICE=1 rank r runs on device r % device_count (all ranks on cuda:0 of a 1-GPU
box), torch.distributed uses gloo (RCCL refuses two ranks on one device) with host tensors for the clock / kernel-time
exchange, every rank's zone walk is bounded to its share of the free memory, and the line is labelled REHEARSAL -- real
engines, real launches, concurrent walks, the whole launcher / barrier / gather flow; not a measurement.
PGX_BENCH_FORCE_DIST=1 initialises the process group even for one rank, so that the RCCL path (init with device_id,
barrier, all_reduce, all_gather on device tensors) runs for real on a 1-GPU box (tests/test_bench_rehearsal_gpu.py).
Timing: W warm-up steps, then `--windows` windows of EXACTLY K steps each, every window bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `value` comes from the MEDIAN window (a single
20-step window is 2.6 ms of GPU time -- one lucky or unlucky sample), all windows are listed in the line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OBS_BYTES = {"float32": 4, "bfloat16": 2, "float16": 2, "uint8": 1}  # bytes per observation cell


def _torch_obs_dtype(name):
    import torch
    return {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16, "uint8": torch.uint8}[name]


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (envs per GPU, size, agents, obs_radius)  -- BASELINE.json configs[1..4]
    "cfg1": (1024, 16, 8, 5),
    "cfg2": (8192, 64, 64, 5),
    "cfg3": (8192, 32, 16, 5),   # configs[3]: 65536 envs over 8 GPUs (`--global-batch 65536`) = 8192 per GPU
    "cfg4": (4096, 256, 256, 7),
}


def algorithmic_bytes_per_agent_step(size: int, agents: int, r: int, obs_bytes: int = 4, action_bytes: int = 1) -> float:
    """12*W^2 obs + 3*ceil(P^2/8)/A bitmaps + 21 bytes of per-agent state/IO, of which
    1 byte is the int8 action (3 * obs_bytes * W^2 for the observation when the non-drop-in uint8 mode is benchmarked;
    wider action dtypes are budgeted at their real width)."""
    W, P = 2 * r + 1, size + 2 * r
    return 3.0 * obs_bytes * W * W + 3.0 * ((P * P + 7) // 8) / agents + 20.0 + action_bytes


# ----------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the oracle is the thing being timed here, never the product path
# ----------------------------------------------------------------------------------------------------------
def cpu_baseline(size, agents, r, collision, density, max_steps, target_seconds=12.0, python_seconds=4.0):
    """Times the plain-C oracle port (oracle/, kind 'port') on the host cores on a bounded sample of the same
    workload, plus the pure-Python literal oracle (closest in spirit to the reference's per-agent Python loops) on a
    smaller sample.  Reported next to the GPU number; never the thing being measured."""
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
    line = {"value": B * agents * steps / dt, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "single_core_value": single, "host_cores_available": avail,
            "sample": f"{B} envs x {agents} agents x {steps} steps of the same workload ({size}x{size}, r={r}, "
                      f"{collision}), plain-C oracle port with OpenMP over envs, {dt:.1f} s wall"}
    # pure-Python literal oracle: one env at a time, dicts and per-agent numpy slices like the reference
    from oracle.pogema_oracle import PogemaOracle
    nb = 2
    penvs = [PogemaOracle(obstacles[b], agents_xy[b], targets_xy[b], obs_radius=r, collision_system=collision,
                          on_target="finish", max_episode_steps=max_steps, auto_reset=True) for b in range(nb)]
    psteps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < python_seconds:
        for b, e in enumerate(penvs):
            e.step(pool[psteps % 16][b])
        psteps += 1
    pdt = time.perf_counter() - t0
    line["python_literal"] = {"value": nb * agents * psteps / pdt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                              "sample": f"{nb} envs x {agents} agents x {psteps} steps, pure-Python literal oracle "
                                        f"(dicts + per-agent numpy slices), {pdt:.1f} s wall"}
    return line


# ----------------------------------------------------------------------------------------------------------
# box fingerprint: what kind of MI355X is this?  (boxes of one pool differ by 25 % in store bandwidth and
# some show no HBM "zones" at all -- the fingerprint lets such boxes be told apart by something other than timing)
# ----------------------------------------------------------------------------------------------------------
def _read(path, limit=4096):
    try:
        with open(path) as f:
            return f.read(limit).strip()
    except OSError:
        return None


def _props(path):
    text = _read(path, 1 << 16)
    if not text:
        return {}
    out = {}
    for ln in text.splitlines():
        k, _, v = ln.partition(" ")
        out[k] = int(v) if v.strip().lstrip("-").isdigit() else v.strip()
    return out


def box_fingerprint():
    """sysfs only (plain file reads: no child process, safe before and after the GPU is initialised): per GPU the
    compute / memory partition modes (SPX/CPX..., NPS1/NPS4...), VRAM vendor and size, VBIOS, KFD node properties (XCCs,
    SIMDs, clocks) and every memory bank (heap type, size, width, clock).  tools/box_probe.sh adds rocm-smi / amd-smi."""
    import glob
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        cards.append({"card": os.path.basename(os.path.dirname(dev)), "device_id": _read(os.path.join(dev, "device")),
                      "unique_id": _read(os.path.join(dev, "unique_id")),
                      "compute_partition": _read(os.path.join(dev, "current_compute_partition")),
                      "memory_partition": _read(os.path.join(dev, "current_memory_partition")),
                      "vram_vendor": _read(os.path.join(dev, "mem_info_vram_vendor")),
                      "vram_total": _read(os.path.join(dev, "mem_info_vram_total")),
                      "vis_vram_total": _read(os.path.join(dev, "mem_info_vis_vram_total")),
                      "vbios_version": _read(os.path.join(dev, "vbios_version")),
                      "pcie_link": f"{_read(os.path.join(dev, 'current_link_speed'))} x{_read(os.path.join(dev, 'current_link_width'))}"})
    nodes = []
    for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*"), key=lambda p: int(os.path.basename(p))):
        pr = _props(os.path.join(node, "properties"))
        if not pr.get("simd_count"):
            continue  # CPU node
        keep = ("gfx_target_version", "simd_count", "array_count", "num_xcc", "cu_per_simd_array", "simd_per_cu",
                "max_engine_clk_fcompute", "local_mem_size", "device_id", "unique_id", "num_sdma_engines", "location_id")
        banks = []
        for bank in sorted(glob.glob(os.path.join(node, "mem_banks", "[0-9]*"))):
            bp = _props(os.path.join(bank, "properties"))
            banks.append({k: bp.get(k) for k in ("heap_type", "size_in_bytes", "flags", "width", "mem_clk_max")})
        nodes.append({"node": int(os.path.basename(node)), **{k: pr.get(k) for k in keep}, "mem_banks": banks})
    # the cards of the HOST are all listed in sysfs; the process sees the KFD nodes its cgroup admits: match by unique id
    mine = {int(n["unique_id"]) for n in nodes if isinstance(n.get("unique_id"), int)}
    for c in cards:
        try:
            c["visible_to_this_process"] = int(c["unique_id"], 16) in mine
        except (TypeError, ValueError):
            c["visible_to_this_process"] = None
    hidden = [c for c in cards if c["visible_to_this_process"] is False]
    if hidden and len(hidden) < len(cards):  # keep the line short: full details of this process's GPUs only
        cards = [c for c in cards if c["visible_to_this_process"] is not False]
        cards.append({"other_cards_on_host": len(hidden), "their_partitions": sorted({f"{c['compute_partition']}/{c['memory_partition']}" for c in hidden}),
                      "their_vram_vendors": sorted({str(c["vram_vendor"]) for c in hidden})})
    return {"drm_cards": cards, "kfd_gpu_nodes": nodes, "kernel": _read("/proc/sys/kernel/osrelease"),
            "amdgpu_version": _read("/sys/module/amdgpu/version"), "hostname": socket.gethostname()}


# ----------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ----------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_device_count() -> int:
    """Number of HIP devices WITHOUT touching the HIP runtime in this process (the launcher must not initialise the GPU:
    its children own the devices).  Round 6: counted from sysfs -- the KFD topology nodes this process's
    cgroup admits that have SIMDs (`box_fingerprint` reads the same files) -- narrowed by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set; only if sysfs shows nothing at all (no KFD: not a ROCm box)
    does it fall back to torch.cuda.device_count(), which merely enumerates on this image."""
    import glob
    n = 0
    for node in glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*"):
        if _props(os.path.join(node, "properties")).get("simd_count"):
            n += 1
    if n == 0:
        import torch
        return int(torch.cuda.device_count())
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def share_device() -> bool:
    """PGX_BENCH_SHARE_DEVICE=1: rehearsal of the N > 1 flow on fewer devices than ranks (module docstring)."""
    return os.environ.get("PGX_BENCH_SHARE_DEVICE", "0") not in ("", "0")


def launch_ranks(n: int, argv, stub: bool = False) -> int:
    """Start `n` rank processes of this script (RANK = LOCAL_RANK = 0..n-1, rendezvous on 127.0.0.1) and wait for
    them.  Returns the worst exit code.  Rank 0's stdout (the JSON line) is passed through."""
    if not stub:
        have = visible_device_count()
        if have < 1 or (have < n and not share_device()):
            print(f"bench.py: --gpus {n} requested but this box exposes {have} HIP device(s); refusing to print a "
                  f"line for fewer GPUs than asked", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PGX_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    worst = 0
    deadline = time.time() + float(os.environ.get("PGX_BENCH_TIMEOUT", "1800"))
    for p in procs:
        try:
            rc = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the child we started
            rc = 124
        worst = max(worst, abs(rc))
    if worst:
        print(f"bench.py: a rank exited with code {worst}", file=sys.stderr)
    return worst


# ----------------------------------------------------------------------------------------------------------
# process groups: gloo is the default group (host side: agreement, per-rank records, HostGather); RCCL carries the clock
# ----------------------------------------------------------------------------------------------------------
def setup_groups(rank, world, device, try_rccl, fake_fail_ranks=()):
    """-> (clock_group or None = the default gloo group, comm_device, label).

    The default process group is ALWAYS gloo: it comes up wherever TCP to 127.0.0.1 works, and every decision that all
    ranks must take together is taken on it.  With `try_rccl` an RCCL group over all ranks is created on top and exercised
    once (all_reduce on a device tensor); whether it is USED is agreed collectively (all_reduce MIN of the per-rank
    verdicts over gloo), so the ranks can never end up on mixed backends and nothing is re-initialised on the
    same port.  A rank that hangs inside RCCL's bootstrap while another one has already failed is ended by a watchdog
    after PGX_BENCH_GROUP_TIMEOUT seconds (default 300) with a non-zero exit code -- loud and bounded instead of the
    launcher's 1800 s."""
    import datetime
    import threading
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:  # PGX_BENCH_FORCE_DIST without a launcher
        os.environ.update(MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    limit = float(os.environ.get("PGX_BENCH_GROUP_TIMEOUT", "300"))

    def give_up():
        print(f"bench.py: rank {rank} did not get its process groups up within {limit:.0f} s; exiting", file=sys.stderr, flush=True)
        os._exit(75)

    dog = threading.Timer(limit, give_up)
    dog.daemon = True
    dog.start()
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if not try_rccl:
            return None, torch.device("cpu"), "gloo"
        group, ok, why = None, 1, ""
        try:
            if rank in fake_fail_ranks:
                raise RuntimeError("PGX_BENCH_FAKE_RCCL_FAIL (test hook)")
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=min(limit, 120.0)),
                                   device_id=device if device.type == "cuda" else None)
            t = torch.ones(1, device=device)
            dist.all_reduce(t, group=group)
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            if int(t.item()) != world:
                raise RuntimeError(f"RCCL all_reduce returned {t.item()} for {world} ranks")
        except Exception as exc:  # noqa: BLE001
            ok, why = 0, repr(exc)
            print(f"bench.py: RCCL is not usable on rank {rank} ({why})", file=sys.stderr, flush=True)
        verdict = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)  # gloo: the collective decision
        if int(verdict.item()) == 1:
            return group, device, "nccl"
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        first = next((f"rank {r}: {w}" for r, w in enumerate(reasons) if w), "unknown")
        return None, torch.device("cpu"), f"gloo (RCCL unusable on {sum(1 for w in reasons if w)} of {world} ranks; {first})"
    finally:
        dog.cancel()


def _pinned_semantics_source():
    """File of reference-pinned semantics defaults in force (tools/pin_reference.sh), or None = the builder's recollections."""
    from pogema_amd.semantics import pinned_source
    return pinned_source()


def _walks_in_process():
    """Full-budget zone walks this process has run, over all its engines (a failed one is paid once: negative cache)."""
    from pogema_amd.buffers import WalkVerdicts
    return {"walks": WalkVerdicts.walks, "failed": WalkVerdicts.failed_walks}


def gpu_identity(dev_index):
    """What tells this rank's GPU apart from its neighbours in the line: HIP uuid (= the unique_id of sysfs / KFD) and
    PCI address."""
    import torch
    out = {"uuid": None, "unique_id": None, "pci": None, "name": None, "total_gib": None}
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        out.update(name=pr.name, total_gib=round(pr.total_memory / float(1 << 30), 1))
        u = getattr(pr, "uuid", None)
        out["uuid"] = str(u) if u is not None else None
        try:  # ROCm fills the uuid with the ASCII of the 16 hex digits that sysfs / KFD / rocm-smi call unique_id (`box` field)
            text = bytes.fromhex(out["uuid"].replace("-", "")).decode("ascii")
            if len(text) == 16 and all(c in "0123456789abcdefABCDEF" for c in text):
                out["unique_id"] = "0x" + text.lower()
        except (ValueError, AttributeError):
            pass
        if hasattr(pr, "pci_bus_id"):
            out["pci"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    except Exception:  # noqa: BLE001
        pass
    return out


# ----------------------------------------------------------------------------------------------------------
# the step under test
# ----------------------------------------------------------------------------------------------------------
def workload_grid_config(args, size, agents, r):
    """The GridConfig every engine of this benchmark is built from."""
    from pogema_amd import GridConfig
    return GridConfig(size=size, density=args.density, num_agents=agents, obs_radius=r, seed=0,
                      collision_system=args.collision, on_target=args.on_target, max_episode_steps=args.max_episode_steps)


def placement_budget(args, world: int = 1):
    """`placement_budget_gib` of every engine of this benchmark.  Default: None = the PRODUCT default (VecPogema walks
    into another HBM zone only when the device is evidently its own, holding half of the free memory meanwhile) -- the
    headline is what a default-constructed engine does.  `--placement-budget all|half|<GiB>` asks explicitly.
    Rehearsal (ranks share a device): an explicit equal share of half of what is free now, so that `world`
    concurrent walks cannot exhaust the device."""
    if share_device() and world > 1:
        import torch
        free, _ = torch.cuda.mem_get_info()
        return max(8.0, 0.5 * free / float(1 << 30) / world)
    b = getattr(args, "placement_budget", "default")
    return None if b == "default" else b if b in ("all", "half") else float(b)


def build_env(args, device, batch, env_base, size, agents, r, placement_probe=True, buffers=None, world=1):
    """The engine whose step() is timed as `value` (also what tests/test_fullsize_gpu.py compares its parity engines'
    launch geometry with)."""
    import torch
    from pogema_amd import VecPogema
    return VecPogema(workload_grid_config(args, size, agents, r), batch=batch, device=device, env_index_base=env_base,
                     auto_reset=True if args.auto_reset == "restore" else "regenerate",
                     # (--graph: the captured steps need fixed buffers -> two alternating sets)
                     reuse_buffers={0: True if args.graph > 0 else "recycle", 1: "single", 2: True}[args.buffers]
                     if buffers is None else buffers,
                     obs_dtype=_torch_obs_dtype(args.obs_dtype),
                     placement_probe=placement_probe,
                     placement_budget_gib=placement_budget(args, world))


class EngineStep:
    """The product path: VecPogema on this rank's device."""

    def __init__(self, args, rank, device, batch, env_base, size, agents, r, placement_probe=True, buffers=None, world=1):
        import torch
        self.torch = torch
        self.env = build_env(args, device, batch, env_base, size, agents, r, placement_probe, buffers, world)
        self.env.reset(seed=0)
        self.env.warm_buffers()  # zone walk + candidate timing + XCD shares here, not inside the first timed step
        tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[args.action_dtype]
        gen = torch.Generator(device=device)
        gen.manual_seed(1 + rank)
        self.pool = [torch.randint(0, 5, (batch, agents), generator=gen, device=device).to(tdt) for _ in range(32)]
        self.no_obs = args.no_obs
        self.nbuf = 2 if (args.buffers == 0 and args.graph > 0) else args.buffers
        self.graph = None
        self.graph_len = 0
        self.i = 0
        if args.graph > 0:
            self._capture(args.graph)

    def _capture(self, n):
        """`--graph n`: n consecutive steps (n kernel launches, alternating output buffers, n different action
        tensors) captured in one HIP graph -- removes the host launch cost between the short kernels of configs[1]."""
        torch = self.torch
        for k in range(4):
            self.env.step(self.pool[k % len(self.pool)], compute_obs=not self.no_obs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(n):
                self.env.step(self.pool[k % len(self.pool)], compute_obs=not self.no_obs)
        self.graph, self.graph_len = g, n

    def run(self, steps):
        if self.graph is not None:
            for _ in range(steps // self.graph_len):
                self.graph.replay()
            return
        for _ in range(steps):
            self.env.step(self.pool[self.i % len(self.pool)], compute_obs=not self.no_obs)
            self.i += 1

    def measure_held_pair(self, steps, windows=3):
        """Secondary figure: the caller keeps step t's outputs alive until step t+1 has returned."""
        torch = self.torch
        rec = self.env._recycler
        out = []
        held = None
        misses0 = rec.misses
        for w in range(windows + 1):  # the first window warms up
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(steps):
                nxt = self.env.step(self.pool[self.i % len(self.pool)])
                held = nxt  # the previous outputs are dropped only now, after the next step has its buffers
                self.i += 1
            ev1.record()
            torch.cuda.synchronize()
            if w:
                out.append(ev0.elapsed_time(ev1) / steps)
        del held, nxt
        return {"ms_per_step": statistics.median(out), "misses": rec.misses - misses0, "output_sets": len(rec)}

    def describe_buffers(self):
        pl = getattr(self.env, "placement", None) or {}
        rec = getattr(self.env, "_recycler", None)
        if rec:
            mode = (f"reuse_buffers='recycle' (product default): {len(rec)} output sets handed out as ordinary tensors and taken "
                    f"back once dropped ({rec.taken} hand-outs, {rec.misses} fell back to fresh tensors); observation buffers")
        else:
            mode = f"{self.nbuf} alternating buffers" if self.nbuf == 2 else "one output set rewritten in place; buffer"
        if pl.get("method", "").startswith("pgx_buffers"):
            return (f"{mode} from the engine's zone-aware pool ({pl.get('pools_tried', 1)} tried): halves in two HBM zones = {pl['spread']} "
                    f"(probe stream {pl['same_zone_us']:.1f} us same-zone -> {pl['final_us']:.1f} us as placed, "
                    f"{pl['candidates']} candidates, {pl['spacer_gib']:.0f} GiB of temporary spacers; plain store stream "
                    f"into the slowest buffer: {pl['buffer_gbs']:.0f} GB/s); observation stream timed into them "
                    f"{pl.get('observe_us_zone')} us vs best of torch's own buffers {pl.get('observe_us_torch_best')} us -> "
                    f"kept {pl.get('chosen')} at {pl.get('observe_us')} us; workgroups per XCD tuned to {pl.get('xcd_shares')}: "
                    f"observation pass {pl.get('observe_us_equal_shares')} -> {pl.get('observe_us_tuned_shares')} us")
        tuned = (f"; workgroups per XCD tuned to {pl['xcd_shares']}: observation pass {pl.get('observe_us_equal_shares')} -> "
                 f"{pl.get('observe_us_tuned_shares')} us") if pl.get("xcd_shares") else ""
        return f"{mode} as torch's allocator returned them (no zone placement){tuned}"

    def placement_fields(self):
        """roofline.placement: where the observation buffers of the timed engine lie, as structured fields -- a line at
        0.68 of peak explains itself (spread false: no second zone found, or no walk under the policy)."""
        pl = getattr(self.env, "placement", None) or {}
        rec = getattr(self.env, "_recycler", None)
        return {"spread": bool(pl.get("spread", False)), "walk_candidates": int(pl.get("candidates", 0) or 0),
                "budget_gib": pl.get("budget_gib"), "spacer_gib_held": pl.get("spacer_gib"),
                "policy": pl.get("policy"), "method": pl.get("method"), "pools_tried": pl.get("pools_tried"),
                "fallback": pl.get("fallback"), "chosen": pl.get("chosen"), "observe_us": pl.get("observe_us"),
                "probe_same_zone_us": pl.get("same_zone_us"), "probe_as_placed_us": pl.get("final_us"),
                "xcd_shares": pl.get("xcd_shares"),
                "output_sets": len(rec) if rec else self.nbuf,
                "recycler": ({"hand_outs": rec.taken, "misses": rec.misses} if rec else None)}

    def box_store_stream_gbs(self):
        """What a bare store stream sustains on THIS box where the buffers lie (the zone walk's probe: 2 x 384 MiB,
        8192 chunks, as placed) -- boxes of the pool differ by 25 % here, the kernel cannot beat it.  None without a walk."""
        pl = getattr(self.env, "placement", None) or {}
        us = pl.get("final_us") or 0.0
        return (2 * (384 << 20) / (us * 1e-6) / 1e9) if us > 0 else None

    def measure_host_gather(self, steps, global_batch, windows=3, with_dist=False):
        """Secondary figure: the host-side gather (north_star "host-side gather only") riding on the step loop.  Every
        step's small outputs -- rewards f32, terminated / truncated / is_active (7 bytes per agent), episode_done and the
        six episode metrics per env -- go to the host through pogema_amd.sharding.HostGather (page-locked segment, async
        D2H on a side stream, finish() of step t-1 while step t runs), against the same loop without it.  Plus the raw
        D2H leg: one observation tensor copied to pinned host memory, GB/s against PCIe Gen5 x16 (63 GB/s spec).
        Collective when `with_dist` (every rank calls it; the shared segment is mapped by all of them)."""
        import torch.distributed as dist
        from pogema_amd.sharding import HostGather, start_step_gather, step_output_fields
        torch, env = self.torch, self.env

        class PeerFailed(RuntimeError):
            pass

        def agree(ok: bool = True) -> bool:
            """Every synchronisation point of this collective figure is ONE all_reduce(MIN) of an ok flag over the default
            (gloo) group: a rank that failed in a phase reports 0 where its peers report 1, they all see 0 at
            the same call and give up together -- with `dist.barrier()` here a failing rank left the others waiting for it
            until the spin / group timeout and the line was delayed or lost."""
            if not with_dist:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def sync_all():
            torch.cuda.synchronize()
            if not agree(True):
                raise PeerFailed("another rank failed in the host-gather figure")

        def plain(n):
            for _ in range(n):
                self.env.step(self.pool[self.i % len(self.pool)])
                self.i += 1

        depth = 2  # steps in flight between start() and finish(): the consumer reads step t-2 while step t is enqueued
        gather, why = None, None
        try:
            gather = HostGather(step_output_fields(env), global_batch, device=env.device, slots=depth + 2,
                                timeout_s=30.0 if with_dist else 120.0)
        except Exception as exc:  # noqa: BLE001
            why = repr(exc)
        if not agree(gather is not None):
            if gather is not None:
                gather.close()
            raise RuntimeError(f"HostGather could not be set up on every rank ({why or 'another rank failed'})")
        fields_bytes = sum(row for _, row in gather._layout.values()) * gather.count
        checked = {}

        def gathered(n):
            pend = []
            for _ in range(n):
                out = self.env.step(self.pool[self.i % len(self.pool)])
                self.i += 1
                pend.append(start_step_gather(gather, out))
                del out
                if len(pend) > depth:
                    host = gather.finish(pend.pop(0))
                    if host is not None and not checked:
                        checked["rewards_sum"] = float(host["rewards"].sum())
            while pend:
                gather.finish(pend.pop(0))

        res = {}
        try:
            for name, fn in (("loop_ms_per_step", plain), ("with_gather_ms_per_step", gathered)):
                fn(max(8, steps // 10))
                out = []
                for _ in range(windows):
                    sync_all()
                    t0 = time.perf_counter()
                    fn(steps)
                    sync_all()
                    out.append((time.perf_counter() - t0) / steps * 1e3)
                res[name] = statistics.median(out)
        except PeerFailed:
            gather.close()
            raise
        except Exception:
            agree(False)  # pairs with the peers' next sync_all(): they stop at the same call instead of waiting for us
            gather.close()
            raise
        res.update(mode=gather.mode, copies_per_step=gather.copies_per_step, bytes_per_step_per_rank=fields_bytes,
                   steps_in_flight=depth, slots=gather.slots,
                   cost_us_per_step=round((res["with_gather_ms_per_step"] - res["loop_ms_per_step"]) * 1e3, 2),
                   small_output_gbs=fields_bytes / (res["with_gather_ms_per_step"] * 1e-3) / 1e9)
        gather.close()
        # the D2H leg on its own: one whole observation tensor into pinned host memory
        obs = env.observe(out=torch.empty(env.obs_shape, dtype=env.obs_dtype, device=env.device))
        nbytes = obs.numel() * obs.element_size()
        if nbytes <= (4 << 30):
            host = torch.empty(obs.shape, dtype=obs.dtype, pin_memory=True)
            side = torch.cuda.Stream(device=env.device)
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                host.copy_(obs, non_blocking=True)
                side.synchronize()
                reps = max(2, min(8, int((2 << 30) // nbytes)))
                t0 = time.perf_counter()
                for _ in range(reps):
                    host.copy_(obs, non_blocking=True)
                side.synchronize()
                dt = time.perf_counter() - t0
            res.update(obs_d2h_gbs=nbytes * reps / dt / 1e9, obs_d2h_ms=dt / reps * 1e3, obs_bytes=nbytes, pcie_spec_gbs=63.0,
                       obs_d2h_frac_of_pcie=nbytes * reps / dt / 1e9 / 63.0,
                       obs_d2h_equal=bool(torch.equal(host, obs.cpu())))
            del host
        return res

    def close(self):
        self.env.close()


class PipelinedStep:
    """Secondary figure: the same batch as TWO engines on two HIP streams, stepped alternately and never joined
    (pogema_amd.PipelinedVecPogema -- double-buffered sampling).  One `step` here = one step of BOTH halves.
    Round 6: with `parent`, the halves write their rows of two FULL-BATCH output sets borrowed from the
    headline engine's recycler (buffers that engine has placed, timed and kept) instead of walking for buffers of their
    own; without, both halves share ONE walk (PipelinedVecPogema._shared_walk) where round 5 ran two."""

    def __init__(self, args, rank, device, batch, env_base, size, agents, r, parts=2, parent=None):
        import torch
        from pogema_amd import GridConfig, PipelinedVecPogema
        self.torch = torch
        gc = GridConfig(size=size, density=args.density, num_agents=agents, obs_radius=r, seed=0,
                        collision_system=args.collision, on_target=args.on_target,
                        max_episode_steps=args.max_episode_steps)
        self.borrowed, parents = None, None
        rec = getattr(parent, "_recycler", None) if parent is not None else None
        if rec and rec.free_sets() >= 2 and tuple(parent.obs_shape)[0] == batch:
            self.borrowed = [rec.take() for _ in range(2)]  # out of circulation for as long as this object lives
            parents = [s[0] for s in self.borrowed]
        self.buffers = "rows of two output sets of the headline engine" if parents else "own buffers, one walk shared by both halves"
        self.placement_spread = bool((getattr(parent, "placement", None) or {}).get("spread", False)) if parents else None
        kw = dict(obs_parents=parents) if parents else dict(reuse_buffers=True, placement_budget_gib=placement_budget(args))
        self.env = PipelinedVecPogema(gc, batch=batch, device=device, parts=parts, env_index_base=env_base, auto_reset=True,
                                      obs_dtype=_torch_obs_dtype(args.obs_dtype), **kw)
        self.env.reset(seed=0)
        self.env.warm_buffers()
        if self.placement_spread is None:
            self.placement_spread = all(bool((e.placement or {}).get("spread", False)) for e in self.env.engines)
        tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[args.action_dtype]
        gen = torch.Generator(device=device)
        gen.manual_seed(1 + rank)
        self.pool = [[torch.randint(0, 5, (batch // parts, agents), generator=gen, device=device).to(tdt) for _ in range(parts)]
                     for _ in range(16)]
        self.env.synchronize()
        torch.cuda.synchronize(device)
        self.i = 0

    def run(self, steps):
        for _ in range(steps):
            self.env.step(self.pool[self.i % len(self.pool)])
            self.i += 1

    def measure(self, steps, windows=3):
        torch = self.torch
        self.run(max(8, steps // 10))
        out = []
        for _ in range(windows):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.run(steps)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps * 1e3)
        return statistics.median(out)

    def close(self):
        self.env.close()
        self.borrowed = None


class RolloutStep:
    """Secondary figure: K steps per launch (pgx_rollout) -- the step kernel's work as one on-device loop (agent / env
    state in registers, no launch boundary, no store drain between steps).  Runs on the engine it is given (round 6: the
    HEADLINE engine, whose rollout() borrows the ring from the output sets that engine has already placed and timed --
    no walk of its own); `slots`: ring size, by default the smallest one of >= 1 GiB (a smaller ring is partly absorbed by
    the 256 MiB Infinity Cache: profiles/r6/rollout_ring_sizes_and_helper_waves.txt)."""

    def __init__(self, args, rank, env, k=64, slots=None):
        import torch
        self.torch, self.k, self.env = torch, k, env
        batch, agents, r = env.batch, env.num_agents, env.obs_radius
        tdt = {"int8": torch.int8, "int32": torch.int32, "int64": torch.int64}[args.action_dtype]
        gen = torch.Generator(device=env.device)
        gen.manual_seed(1 + rank)
        self.actions = torch.randint(0, 5, (k, batch, agents), generator=gen, device=env.device).to(tdt)
        self.obs_bytes = batch * agents * 3 * (2 * r + 1) ** 2 * OBS_BYTES[args.obs_dtype]
        self.slots = slots if slots is not None else min(k, max(2, -(-(1 << 30) // self.obs_bytes)))
        self.ring = None

    def measure(self, steps, windows=3):
        torch = self.torch
        launches = max(1, steps // self.k)
        out = self.env.rollout(self.actions, obs_slots=self.slots)
        rec = getattr(self.env, "_recycler", None)
        ptrs = set(rec.obs_pointers()) if rec else set()
        self.ring = ("output sets of this engine (recycler)" if out["obs"] is not None and out["obs"].data_ptr() in ptrs
                     else "zone pool of its own" if self.slots in getattr(self.env, "_rollout_pools", {}) and
                     self.env._rollout_pools[self.slots] is not None else "torch allocator (dense)")
        del out
        res = []
        for _ in range(windows):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(launches):
                self.env.rollout(self.actions, obs_slots=self.slots)
            ev1.record()
            torch.cuda.synchronize()
            res.append(ev0.elapsed_time(ev1) / (launches * self.k))
        return statistics.median(res)

    def fields(self, ms, alg_bytes):
        ring_mib = self.slots * self.obs_bytes / float(1 << 20)
        return {"steps_per_launch": self.k, "obs_slots": self.slots, "ring_mib": round(ring_mib, 1), "ring_memory": self.ring,
                "ring_vs_infinity_cache": "HBM-sized (>= 1 GiB)" if ring_mib >= 1024 else
                                          "within reach of the 256 MiB Infinity Cache: NOT an HBM figure",
                "ms_per_step": ms, "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "placement_spread": bool((self.env.placement or {}).get("spread", False))}


class StubStep:
    """TEST ONLY (`--stub`): no GPU, no engine -- a fixed host sleep per step so that tests/test_bench_launch.py can
    drive the launcher, the barriers, the max-over-ranks reduction and the JSON contract on CPU over gloo.  A line
    produced this way says data = "stub" and is not a measurement."""

    def __init__(self, rank):
        self.delay = 0.0005 * (1 + rank)

    def run(self, steps):
        time.sleep(self.delay * steps)

    def describe_buffers(self):
        return "stub"

    def close(self):
        pass


def headline_form(args, batch, per_gpu) -> bool:
    """The default invocation (what the driver runs): the headline workload as the product runs it."""
    return (args.workload == "cfg2" and args.obs_dtype == "float32" and batch == per_gpu and args.global_batch == 0
            and args.buffers == 0 and args.graph == 0 and not args.no_obs and args.auto_reset == "restore")


def measure_workload(args, name, rank, device):
    """One more BASELINE config inside the headline process: a default-constructed engine of `name` (its own buffers: the
    recycling allocator, a zone walk where the observation tensor is large enough to deserve one -- the line says what
    it got), timed with HIP events over its own step() loop, then its rollout (ring borrowed from the engine's output
    sets) and, for launches too short to hide the host, 32 steps per HIP graph."""
    import copy
    import torch
    per_gpu, size, agents, r = WORKLOADS[name]
    wargs = copy.copy(args)
    wargs.workload, wargs.graph, wargs.buffers = name, 0, 0
    bpas = algorithmic_bytes_per_agent_step(size, agents, r, OBS_BYTES[args.obs_dtype], {"int8": 1, "int32": 4, "int64": 8}[args.action_dtype])
    alg_bytes = bpas * per_gpu * agents
    obs_bytes = per_gpu * agents * 3 * (2 * r + 1) ** 2 * OBS_BYTES[args.obs_dtype]
    n = 400 if obs_bytes < (64 << 20) else 200 if obs_bytes < (1 << 30) else 50
    t_build = time.perf_counter()
    st = EngineStep(wargs, rank, device, per_gpu, 0, size, agents, r)
    build_s = time.perf_counter() - t_build
    out = {"config": f"BASELINE.json configs[{name[-1]}]: {per_gpu} envs, {size}x{size} map, {agents} agents, obs_radius {r}",
           "steps_timed": n, "algorithmic_bytes_per_launch": alg_bytes, "engine_build_s": round(build_s, 2)}
    try:
        st.run(max(10, n // 10))
        ms = []
        for _ in range(3):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            st.run(n)
            ev1.record()
            torch.cuda.synchronize()
            ms.append(ev0.elapsed_time(ev1) / n)
        step_ms = statistics.median(ms)
        gbs = alg_bytes / (step_ms * 1e-3) / 1e9
        box = st.box_store_stream_gbs()
        pf = st.placement_fields()
        geo = st.env.geometry()
        out.update(ms_per_step=step_ms, value=per_gpu * agents / (step_ms * 1e-3), unit="agent-steps/s", achieved_gbs=gbs,
                   frac=gbs / HBM_PEAK_GBS, box_store_stream_gbs=box, frac_of_box_store_stream=(gbs / box) if box else None,
                   placement_spread=pf["spread"], placement_policy=pf["policy"], output_sets=pf["output_sets"],
                   geometry={k: geo[k] for k in ("lanes_per_env", "waves", "envs_per_wave", "multi_wave", "p16", "grid", "lds_bytes") if k in geo})
        rs = RolloutStep(wargs, rank, st.env, slots=2)
        out["rollout"] = rs.fields(rs.measure(max(64, n)), alg_bytes)
        out["rollout_ms_per_step"] = out["rollout"]["ms_per_step"]
        out["rollout"]["geometry"] = {k: v for k, v in st.env.geometry(for_rollout=True).items()
                                      if k in ("lanes_per_env", "waves", "envs_per_wave", "grid", "lds_bytes")}
        if obs_bytes * 2 < (1 << 30):  # ... and into a ring that the Infinity Cache cannot hold: dense torch memory (no walk)
            probe, st.env.placement_probe = st.env.placement_probe, False
            try:
                rh = RolloutStep(wargs, rank, st.env)
                out["rollout_hbm_ring"] = rh.fields(rh.measure(max(64, n)), alg_bytes)
            finally:
                st.env.placement_probe = probe
            del rh
        del rs
    finally:
        st.close()
        del st
        torch.cuda.empty_cache()
    if obs_bytes < (256 << 20):
        gargs = copy.copy(wargs)
        gargs.graph = 32
        gs = EngineStep(gargs, rank, device, per_gpu, 0, size, agents, r)
        try:
            reps = max(1, n // 32)
            gs.run(64)
            g_ms = []
            for _ in range(3):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record()
                gs.run(32 * reps)
                ev1.record()
                torch.cuda.synchronize()
                g_ms.append(ev0.elapsed_time(ev1) / (32 * reps))
            out["graph_ms_per_step"] = statistics.median(g_ms)
        finally:
            gs.close()
            del gs
    return out


WORKLOAD_KEYS = ("config", "ms_per_step", "value", "frac", "frac_of_box_store_stream", "rollout_ms_per_step", "rollout",
                 "geometry", "placement_spread")  # + graph_ms_per_step for the short launches (cfg1, cfg3)


def stub_workload(name):
    """TEST ONLY (`--stub`): the record of measure_workload with host sleeps instead of engines, so that the CPU suite can
    pin the key set of secondary.workloads (tests/test_bench_launch.py)."""
    per_gpu, size, agents, r = WORKLOADS[name]
    st = StubStep(0)
    t0 = time.perf_counter()
    st.run(4)
    ms = (time.perf_counter() - t0) / 4 * 1e3
    alg = algorithmic_bytes_per_agent_step(size, agents, r) * per_gpu * agents
    rec = {"config": f"STUB configs[{name[-1]}]", "ms_per_step": ms, "value": per_gpu * agents / (ms * 1e-3), "unit": "agent-steps/s",
           "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_box_store_stream": None, "placement_spread": False,
           "geometry": {}, "rollout": {"ms_per_step": ms, "obs_slots": 2}, "rollout_ms_per_step": ms}
    if per_gpu * agents * 3 * (2 * r + 1) ** 2 * 4 < (256 << 20):
        rec["graph_ms_per_step"] = ms
    return rec


def configs0_cpu_baseline(seconds=2.0):
    """BASELINE.json configs[0] -- GridConfig(size=8, num_agents=2, obs_radius=3, density=0.3), ONE env, the reference's own
    CPU-runnable case -- on the pure-Python literal oracle (the closest thing to the reference's step() that exists here)
    and on the plain-C port, one core each."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import generate_instances
    from oracle.pogema_oracle import PogemaOracle
    from oracle.c_oracle import COracle
    size, agents, r = 8, 2, 3
    obstacles, agents_xy, targets_xy = generate_instances(1, size, size, agents, 0.3, 0)
    pool = np.random.default_rng(1).integers(0, 5, size=(64, 1, agents)).astype(np.int64)
    env = PogemaOracle(obstacles[0], agents_xy[0], targets_xy[0], obs_radius=r, collision_system="priority", on_target="finish",
                       max_episode_steps=64, auto_reset=True)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.step(pool[n % 64][0])
        n += 1
    dt = time.perf_counter() - t0
    res = {"value": agents * n / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port", "env_steps_per_s": n / dt,
           "sample": f"1 env x {agents} agents x {n} steps, GridConfig(size=8, num_agents=2, obs_radius=3, density=0.3), "
                     f"pure-Python literal oracle, {dt:.1f} s wall"}
    c = COracle(1, size, size, agents, r, "priority", "finish", 64, True)
    c.reset(obstacles, agents_xy, targets_xy)
    W = 2 * r + 1
    out = (np.empty((1, agents, 3, W, W), np.float32), np.empty((1, agents), np.float32), np.empty((1, agents), np.uint8),
           np.empty((1, agents), np.uint8), np.empty((1, agents), np.uint8))
    m, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        c.step(pool[m % 64], nthreads=1, out=out)
        m += 1
    cdt = time.perf_counter() - t0
    c.close()
    res["c_port"] = {"value": agents * m / cdt, "unit": "agent-steps/s", "cores": 1, "kind": "port", "env_steps_per_s": m / cdt,
                     "sample": f"1 env x {agents} agents x {m} steps, plain-C oracle port through ctypes (call overhead included)"}
    return res


def make_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of exactly --steps steps; value = median window")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--collision", default="soft", choices=["priority", "block_both", "soft"])
    ap.add_argument("--on-target", default="finish", choices=["finish", "restart", "nothing"])
    ap.add_argument("--batch", type=int, default=0, help="override envs per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total envs over all ranks, sharded with sharding.shard_bounds (strong scaling); "
                         "configs[3] = --workload cfg3 --global-batch 65536 on 8 GPUs")
    ap.add_argument("--density", type=float, default=0.3)
    ap.add_argument("--max-episode-steps", type=int, default=64)
    ap.add_argument("--action-dtype", default="int8", choices=["int8", "int32", "int64"],
                    help="int8 = the 1 byte/agent of SURVEY 8(d)'s formula; wider dtypes are budgeted at their width")
    ap.add_argument("--obs-dtype", default="float32", choices=sorted(OBS_BYTES),
                    help="float32 = the reference's dtype (the headline); bfloat16 / float16 / uint8 = the engine's lighter "
                         "non-drop-in formats (the same 0/1 planes in 2 / 2 / 1 bytes per cell)")
    ap.add_argument("--auto-reset", default="restore", choices=["restore", "regenerate"],
                    help="restore = finished envs return to their initial state inside the step kernel (headline); "
                         "regenerate = they get a fresh random instance on the device (pgx_regenerate)")
    ap.add_argument("--buffers", type=int, default=0, choices=[0, 1, 2],
                    help="output buffers: 0 = the product default (reuse_buffers='recycle': ordinary tensors from the engine's "
                         "zone-spread pool, taken back when the caller has dropped them -- the loop drops each step's outputs "
                         "before the next step, as a policy loop does); 2 = two alternating sets (step t's tensors are "
                         "overwritten by step t+2); 1 = one set rewritten in place")
    ap.add_argument("--placement-budget", default="default",
                    help="HBM the engine's zone walk may hold for its 1-3 s: default = the product default (walk only on a "
                         "device that is evidently ours, half of the free memory); half | all | a number of GiB = explicit")
    ap.add_argument("--graph", type=int, default=0, help="capture this many steps in one HIP graph and replay it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-placement", action="store_true",
                    help="skip the extra window that times the unprobed (default allocator) buffers")
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip the observation write")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures (two pipelined engines; K-step rollout launches)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # tests only, see StubStep
    ap.add_argument("--stub-rccl", action="store_true", help=argparse.SUPPRESS)  # tests only: attempt RCCL in stub mode
    return ap


def main(argv=None):
    args = make_parser().parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.windows < 1:
        raise SystemExit("--gpus, --steps and --windows must be >= 1")
    if args.graph > 0 and args.steps % args.graph:
        raise SystemExit(f"--steps {args.steps} must be a multiple of --graph {args.graph}")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None:
        if args.gpus > 1:  # launcher role: start the ranks before anything touches the GPU, pass the verdict on
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv), stub=args.stub))
        world, rank, local_rank = 1, 0, 0
    else:
        world, rank, local_rank = int(world_env), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; they must agree")

    fingerprint = box_fingerprint() if rank == 0 else None  # plain sysfs reads, before anything touches the GPU

    import torch
    import torch.distributed as dist

    fake_fail = tuple(int(x) for x in os.environ.get("PGX_BENCH_FAKE_RCCL_FAIL", "").split(",") if x.strip())
    force_dist = os.environ.get("PGX_BENCH_FORCE_DIST", "0") not in ("", "0")
    clock_group, comm_device, group_label, rehearsal, dev_index = None, torch.device("cpu"), None, False, None
    if args.stub:
        device = torch.device("cpu")
        if world > 1 or force_dist:
            # (--stub-rccl: tests/test_bench_launch.py drives the RCCL attempt and the collective fallback on a box
            # without any GPU -- RCCL fails on every rank there, or on the faked ones first)
            clock_group, comm_device, group_label = setup_groups(rank, world, device, args.stub_rccl, fake_fail)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
        rehearsal = share_device() and world > torch.cuda.device_count()
        if local_rank >= torch.cuda.device_count() and not rehearsal:
            raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, the box has {torch.cuda.device_count()}")
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        if world > 1 or force_dist:
            # rehearsal: RCCL refuses two ranks on one device -> gloo only, host tensors for the clock
            clock_group, comm_device, group_label = setup_groups(rank, world, device, not rehearsal, fake_fail)
    use_dist = dist.is_initialized()

    from pogema_amd.sharding import shard_bounds
    per_gpu, size, agents, r = WORKLOADS[args.workload]
    if args.global_batch > 0:
        env_base, batch = shard_bounds(args.global_batch, world, rank)
        total_envs, scaling = args.global_batch, "strong"
    else:
        batch = args.batch if args.batch > 0 else per_gpu
        env_base, total_envs, scaling = rank * batch, world * batch, "weak"
    if batch < 1:
        raise SystemExit(f"rank {rank} holds no environment (global batch {args.global_batch} over {world} ranks)")

    def sync():
        if not args.stub:
            torch.cuda.synchronize(device)

    def barrier():
        sync()
        if use_dist:
            dist.barrier(group=clock_group)
        sync()

    def reduce_max(x: float) -> float:
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=clock_group)
        return float(t.item())

    def timed_windows(step, n_windows):
        """-> (wall seconds per window, max over ranks; this rank's stream-side ms per step per window)"""
        walls, kernel = [], []
        for _ in range(n_windows):
            barrier()
            if not args.stub:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()  # the (current) stream the engine launches on
            t0 = time.perf_counter()
            step.run(args.steps)
            if not args.stub:
                ev1.record()
            own = time.perf_counter() - t0  # stub only: this rank's own time, before the closing barrier
            barrier()
            dt = time.perf_counter() - t0
            walls.append(reduce_max(dt))
            kernel.append(ev0.elapsed_time(ev1) / args.steps if not args.stub else own / args.steps * 1e3)
        return walls, kernel

    default_ms = None
    step = StubStep(rank) if args.stub else EngineStep(args, rank, device, batch, env_base, size, agents, r, world=world)
    warm = args.warmup if args.graph <= 0 else -(-args.warmup // args.graph) * args.graph
    if warm:
        step.run(warm)
    walls, kernel = timed_windows(step, args.windows)
    elapsed = statistics.median(walls)
    kernel_ms = statistics.median(kernel)
    per_rank_kernel = [kernel_ms]
    if use_dist:
        t = torch.tensor([kernel_ms], dtype=torch.float64, device=comm_device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=clock_group)
        per_rank_kernel = [float(p.item()) for p in parts]

    # every rank's own record: which GPU, where its buffers lie, what a bare store stream does there --
    # with max-over-ranks timing ONE slow GPU sets the node's clock, and the line must say which one and why
    abytes = {"int8": 1, "int32": 4, "int64": 8}[args.action_dtype]
    bpas = algorithmic_bytes_per_agent_step(size, agents, r, OBS_BYTES[args.obs_dtype], abytes)
    my = {"rank": rank, "envs": batch, "kernel_ms": kernel_ms, "host": socket.gethostname()}
    if not args.stub:
        pf = step.placement_fields()
        my_gbs = bpas * batch * agents / (kernel_ms * 1e-3) / 1e9
        box_gbs = step.box_store_stream_gbs()
        my.update(device=dev_index, gpu=gpu_identity(dev_index), achieved_gbs=my_gbs, frac=my_gbs / HBM_PEAK_GBS,
                  spread=pf["spread"], walk_candidates=pf["walk_candidates"], spacer_gib_held=pf["spacer_gib_held"],
                  probe_same_zone_us=pf["probe_same_zone_us"], probe_as_placed_us=pf["probe_as_placed_us"],
                  box_store_stream_gbs=box_gbs, frac_of_box_store_stream=(my_gbs / box_gbs) if box_gbs else None,
                  chosen=pf["chosen"], policy=pf["policy"], fallback=pf["fallback"])
    per_rank = [my]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, my)  # the default group: gloo, host side

    # the same workload into buffers as torch's allocator hands them out (no zone placement), AFTER the main measurement:
    # allocating and freeing them first would leave holes that the zone walk of the main run falls into
    if not args.stub and not args.no_default_placement and world == 1:
        plain = EngineStep(args, rank, device, batch, env_base, size, agents, r, placement_probe=False)
        plain.run(max(args.warmup, 4))
        _, k = timed_windows(plain, 1)
        default_ms = k[0]
        plain.close()
        del plain
        torch.cuda.empty_cache()

    # secondary figures, never `value`: the same work without the launch boundary in the way
    extras = {}
    if (not args.stub and not args.no_extras and world == 1 and args.graph <= 0 and args.auto_reset == "restore"
            and not args.no_obs and args.buffers != 1):
        n = min(args.steps, 1000)
        extra_errors = {}
        if args.buffers == 0 and getattr(step.env, "_recycler", None):
            try:
                extras["held_pair"] = step.measure_held_pair(n)
            except Exception as exc:  # noqa: BLE001
                extra_errors["held_pair"] = repr(exc)
        if batch % 2 == 0:
            try:  # a secondary figure must never cost the headline line
                # (a) the halves write their rows of two full-batch output sets of the headline engine: no walk, no verdict
                # of their own -- but each half's rows lie in ONE zone of such a buffer; (b) buffers of their own, placed
                # by ONE walk shared by both halves (PipelinedVecPogema._shared_walk)
                ps = PipelinedStep(args, rank, device, batch, env_base, size, agents, r, parent=step.env)
                rows = {"ms_per_step": ps.measure(n), "buffers": ps.buffers, "placement_spread": ps.placement_spread}
                ps.close()
                del ps
                torch.cuda.empty_cache()
                ps = PipelinedStep(args, rank, device, batch, env_base, size, agents, r)
                extras["pipelined"] = {"engines": 2, "ms_per_step": ps.measure(n), "buffers": ps.buffers,
                                       "placement_spread": ps.placement_spread, "rows_of_headline_sets": rows}
                ps.close()
                del ps
            except Exception as exc:  # noqa: BLE001
                extra_errors["pipelined"] = repr(exc)
            torch.cuda.empty_cache()
        try:
            rs = RolloutStep(args, rank, step.env)
            extras["rollout"] = rs.fields(rs.measure(n), bpas * batch * agents)
            del rs
        except Exception as exc:  # noqa: BLE001
            extra_errors["rollout"] = repr(exc)
        torch.cuda.empty_cache()
        obs_bytes_step = batch * agents * 3 * (2 * r + 1) ** 2 * OBS_BYTES[args.obs_dtype]
        if obs_bytes_step < (256 << 20):
            # short launches: the same step() calls captured 32 at a time in a HIP graph and replayed -- no Python, no
            # per-launch host work between two kernels: separates the host-bound part of `value` from the kernel
            try:
                import copy
                gargs = copy.copy(args)
                gargs.graph, gargs.buffers = 32, 0
                gs = EngineStep(gargs, rank, device, batch, env_base, size, agents, r, world=world)
                reps = max(1, n // 32)
                gs.run(32 * 2)
                g_ms = []
                for _ in range(3):
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    ev0.record()
                    gs.run(32 * reps)
                    ev1.record()
                    torch.cuda.synchronize()
                    g_ms.append(ev0.elapsed_time(ev1) / (32 * reps))
                extras["graph"] = {"steps_per_graph": 32, "ms_per_step": statistics.median(g_ms)}
                gs.close()
                del gs
            except Exception as exc:  # noqa: BLE001
                extra_errors["graph"] = repr(exc)
            torch.cuda.empty_cache()
        try:
            extras["host_gather"] = step.measure_host_gather(min(n, 400), total_envs)
        except Exception as exc:  # noqa: BLE001
            extra_errors["host_gather"] = repr(exc)
        # every other BASELINE GPU config in the driver's one line: default-constructed engines of
        # configs[1], the configs[3] shard and configs[4] in this same process, each with its step() loop, its rollout and
        # (short launches) its HIP-graph figure; a failure lands in `errors`, never costs the line
        if headline_form(args, batch, per_gpu):
            extras["workloads"] = {}
            for name in ("cfg1", "cfg3", "cfg4"):
                if name == args.workload:
                    continue
                try:
                    extras["workloads"][name] = measure_workload(args, name, rank, device)
                except Exception as exc:  # noqa: BLE001
                    extra_errors["workloads." + name] = repr(exc)
                torch.cuda.empty_cache()
        if extra_errors:
            extras["errors"] = extra_errors
    elif args.stub and not args.no_extras and world == 1 and headline_form(args, batch, per_gpu):
        extras["workloads"] = {name: stub_workload(name) for name in ("cfg1", "cfg3", "cfg4")}
    elif not args.stub and not args.no_extras and world > 1 and args.graph <= 0 and not args.no_obs:
        # N > 1: the gather is the one cross-rank piece of the product path -- every rank takes part (shared segment).  A
        # failure on any rank must not cost the line: inside measure_host_gather every synchronisation point is an
        # all_reduce(MIN) of an ok flag over gloo, so all ranks leave the figure at the same call
        try:
            hg = step.measure_host_gather(min(args.steps, 200), total_envs, with_dist=True)
        except Exception as exc:  # noqa: BLE001
            hg = {"error": repr(exc)}
        if rehearsal:
            hg["rehearsal"] = (f"{world} ranks time-share ONE device, one copy engine set and one PCIe link here: the figures show "
                               f"that the shared-segment gather runs, not what it costs on {world} GPUs")
        extras["host_gather"] = hg

    if rank == 0:
        n_agent_steps = total_envs * agents * args.steps
        value = n_agent_steps / elapsed
        alg_bytes = bpas * batch * agents  # per launch (this GPU's shard)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_file = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    key = f"{args.workload}/{args.collision}" + ("" if args.obs_dtype == "float32" else "/" + args.obs_dtype)
                    entry = json.load(f).get(key, {})
                    traffic, traffic_file = entry.get("hbm_bytes_per_launch"), entry.get("source")
            except Exception:
                traffic = None
        # BASELINE.json's metric label only for BASELINE.json's workload as the product runs it: configs[2], float32,
        # 8192 envs on every GPU, one launch per step, observations written, the DEFAULT output allocator
        # (reuse_buffers='recycle': three output sets handed out in turn; this loop drops every step's outputs before the
        # next step, so no hand-out ever misses -- roofline.placement.recycler; the caller who HOLDS (obs, next_obs) is
        # the secondary figure `held_pair`) -- ADVICE r2 / r3
        headline = (args.workload == "cfg2" and args.obs_dtype == "float32" and not args.stub and batch == per_gpu
                    and args.global_batch == 0 and args.buffers == 0 and args.graph == 0 and not args.no_obs
                    and not rehearsal)
        variant = "".join([f", {batch} envs per GPU" if args.global_batch == 0 else f", global batch {total_envs}",
                           {0: "", 1: ", one output set rewritten in place", 2: ", two alternating output sets"}[args.buffers],
                           f", hipGraph of {args.graph} steps" if args.graph > 0 else "",
                           ", NO observation write (diagnostic)" if args.no_obs else ""])
        line = {
            "metric": "agent-steps/sec (whole node), 64-agent 64x64 grid, batch=8192 envs" if headline
                      else f"{'STUB (not a measurement) ' if args.stub else ''}"
                           f"{f'REHEARSAL ({world} ranks share {torch.cuda.device_count()} device(s); not a measurement) ' if rehearsal else ''}"
                           f"agent-steps/sec (whole node), "
                           f"workload {args.workload}, obs {args.obs_dtype}{variant}",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32", "data": "stub" if args.stub else "synthetic",
            "windows": {"n": args.windows, "statistic": "median", "steps_each": args.steps,
                        "ms_per_step": [w / args.steps * 1e3 for w in walls]},
            "config": {"workload": f"BASELINE.json configs[{args.workload[-1]}]: {total_envs} envs over {world} GPU(s) "
                                   f"({batch} on rank 0), {size}x{size} map, {agents} agents, obs_radius {r}, "
                                   f"density {args.density}",
                       "collision_system": args.collision, "on_target": args.on_target, "auto_reset": args.auto_reset,
                       "max_episode_steps": args.max_episode_steps, "obs_dtype": args.obs_dtype,
                       "action_dtype": args.action_dtype, "envs_per_gpu": batch, "global_batch": total_envs,
                       "sharding": f"batch-sharded x{world}, no collective",
                       "semantics": None if args.stub else {k: getattr(step.env.semantics, k) for k in
                                                                ("soft_vertex", "soft_occupancy", "coop_reward", "bad_action",
                                                                 "lifelong_rng", "generator_rng")},
                       "semantics_pinned_by": None if args.stub else _pinned_semantics_source(),
                       "rehearsal": bool(rehearsal),
                       "process_group": group_label if use_dist else None,
                       "launch": f"hipGraph of {args.graph} steps" if args.graph > 0 else "one pgx_step launch per step",
                       "obs_buffers": step.describe_buffers()},
            "roofline": {"bound": "hbm" if (args.buffers != 1 or alg_bytes > (200 << 20)) else "hbm (output tensor rewritten in place: largely Infinity-Cache resident)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (None if traffic is None else
                                            f"profiles/pmc_traffic.json <- {traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                            "passes of an EARLIER run of this workload (calibrated per MI355X_MICROARCH.md) -- a "
                                            "looked-up figure, not measured by this process"),
                         "peak_note": "8000 GB/s = HBM3E spec (MI355X_MICROARCH.md); the guide's 6.29 TB/s is a float4 COPY "
                                      "(read + write); this kernel is a write-only stream, and bare store streams reach "
                                      "6.7-7.0 TB/s on this pool (DESIGN.md section 6, roofline.box_store_stream_gbs)",
                         "placement": None if args.stub else step.placement_fields(),
                         "kernel": "pgx::step_kernel", "kernel_ms": kernel_ms, "kernel_ms_per_rank": per_rank_kernel,
                         "kernel_ms_windows": kernel,
                         "default_placement_kernel_ms": default_ms,
                         "box_store_stream_gbs": None if args.stub else step.box_store_stream_gbs(),
                         # kernel quality separated from the box: the step against the bare store stream timed on the
                         # same GPU, same placement (1.0 = the kernel IS the box's write ceiling)
                         "frac_of_box_store_stream": per_rank[0].get("frac_of_box_store_stream"),
                         "per_rank": per_rank,
                         "slowest_rank": max(range(len(per_rank)), key=lambda i: per_rank[i]["kernel_ms"]),
                         "zone_walks_in_process": None if args.stub else _walks_in_process(),
                         "algorithmic_bytes_per_agent_step": bpas, "algorithmic_bytes_per_launch": alg_bytes,
                         "profile_command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-default-placement "
                                            "--no-cpu-baseline --no-extras  (the default-placement window and the secondary "
                                            "figures launch the same kernel on other buffers / half batches; profiles/r6/README.md)"},
        }
        if "pipelined" in extras:
            e = extras["pipelined"]
            e.update(value=batch * agents / (e["ms_per_step"] * 1e-3), unit="agent-steps/s",
                     frac=alg_bytes / (e["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     what="the same batch as two engines on two HIP streams, stepped alternately and never joined "
                          "(PipelinedVecPogema, double-buffered sampling): one half's launch boundary lies under the other "
                          "half's observation stream; wall clock over both halves")
        if "rollout" in extras:
            e = extras["rollout"]
            e.update(value=batch * agents / (e["ms_per_step"] * 1e-3), unit="agent-steps/s",
                     what="pgx_rollout on the headline engine: 64 steps per launch with the actions given up front -- one "
                          "on-device loop, agent / env state in registers, nothing waits for a store -- observations into a "
                          "ring of obs_slots tensors (>= 1 GiB in total) borrowed from that engine's own output sets; "
                          "bit-identical with 64 pgx_step calls; HIP events around the launches")
        if "graph" in extras:
            e = extras["graph"]
            e.update(value=batch * agents / (e["ms_per_step"] * 1e-3), unit="agent-steps/s",
                     frac=alg_bytes / (e["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     host_us_per_step_in_value=round((elapsed / args.steps - e["ms_per_step"] * 1e-3) * 1e6, 2),
                     what="the same pgx_step launches, 32 consecutive steps captured in one HIP graph (two alternating output "
                          "sets, 32 different action tensors) and replayed: what the GPU needs per step when no host work "
                          "lies between two launches; `value` minus this is the host-bound part of a Python step() loop")
        if "host_gather" in extras and "error" not in extras["host_gather"]:
            e = extras["host_gather"]
            e.update(what="the step loop with the host-side gather riding on it (pogema_amd.sharding.HostGather: every step's "
                          "rewards / terminated / truncated / is_active / episode_done / metrics into one page-locked host "
                          "segment by async D2H on a side stream, issued by the host once the step's event has completed -- no "
                          "GPU-side cross-stream wait --, finish(t-2) while step t is enqueued; N > 1: all ranks DMA into one "
                          "shared segment, landed / released counters in its header, no collective and no gloo call per step) "
                          "against the same loop without it; obs_d2h_*: one observation tensor to pinned host memory, against PCIe Gen5 x16")
        if "held_pair" in extras:
            e = extras["held_pair"]
            e.update(value=batch * agents / (e["ms_per_step"] * 1e-3), unit="agent-steps/s",
                     frac=alg_bytes / (e["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     what="the headline engine stepped by a caller that HOLDS the previous step's outputs while taking the "
                          "next (obs, next_obs pairs, as a replay-buffer writer does): two of the three recycled output sets "
                          "are out at any time; `misses` = hand-outs that fell back to fresh torch tensors")
        if extras:
            line["secondary"] = extras
        line["box"] = fingerprint
        if world == 1 and not args.no_cpu_baseline and not args.stub:
            try:  # the baseline leg must never cost the measured line
                line["cpu_baseline"] = cpu_baseline(size, agents, r, args.collision, args.density, args.max_episode_steps,
                                                    args.cpu_seconds)
                try:
                    line["cpu_baseline"]["configs0_python_literal"] = configs0_cpu_baseline()
                except Exception as exc:  # noqa: BLE001
                    line["cpu_baseline"]["configs0_python_literal"] = {"value": None, "sample": f"FAILED: {exc!r}"}
            except Exception as exc:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port",
                                        "sample": f"FAILED: {exc!r}"}
        elif world == 1 and not args.no_cpu_baseline and args.stub:  # (tests: the configs[0] leg runs on the CPU for real)
            line["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port", "sample": "stub",
                                    "configs0_python_literal": configs0_cpu_baseline(0.3)}
        print(json.dumps(line), flush=True)
    step.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
