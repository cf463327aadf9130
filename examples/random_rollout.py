#!/usr/bin/env python3
"""Minimal tour of the engine: 4096 environments, a random policy that lives on the GPU, episode metrics collected on
the device, fresh maps for finished environments without a host round trip.

    python examples/random_rollout.py [--envs 4096] [--agents 32] [--size 32] [--steps 512]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pogema_amd import GridConfig, VecPogema  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=32)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--collision", default="soft")
    args = ap.parse_args()

    gc = GridConfig(size=args.size, num_agents=args.agents, obs_radius=5, density=0.3, seed=0,
                    collision_system=args.collision, on_target="finish", max_episode_steps=128)
    env = VecPogema(gc, batch=args.envs, auto_reset="regenerate", reuse_buffers=True)
    obs, infos = env.reset(seed=0)                      # instances are drawn on the GPU
    print("obs", tuple(obs.shape), obs.dtype, "on", obs.device)

    episodes = torch.zeros((), dtype=torch.int64, device=env.device)   # everything stays on the device: no per-step sync
    isr_sum = torch.zeros((), device=env.device)
    for _ in range(8):  # warm-up: torch's random/reduction kernels load lazily, the engine probes its output buffers
        _, _, _, _, infos = env.step(torch.randint(0, 5, (args.envs, args.agents), device=env.device))
        _ = infos["episode_done"].sum() + (infos["metrics"][:, 0] * infos["episode_done"]).sum()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        actions = torch.randint(0, 5, (args.envs, args.agents), device=env.device)   # your policy(obs) goes here
        obs, rewards, terminated, truncated, infos = env.step(actions)
        done = infos["episode_done"]                   # bool [envs]; metrics rows are valid where it is set
        episodes += done.sum()
        isr_sum += (infos["metrics"][:, 0] * done).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.steps} steps x {args.envs} envs x {args.agents} agents in {dt:.3f} s "
          f"= {args.steps * args.envs * args.agents / dt:.3e} agent-steps/s (policy sampling and metric reads included)")
    episodes = int(episodes)
    print(f"{episodes} episodes finished, mean ISR of a random policy: {float(isr_sum) / max(episodes, 1):.3f}; "
          f"environments that could not be regenerated: {env.regenerate_failures()}")
    env.close()


if __name__ == "__main__":
    main()
