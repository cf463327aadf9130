#!/usr/bin/env python3
"""Double-buffered sampling: the batch as two engines on two HIP streams (pogema_amd.PipelinedVecPogema).  Each half's
policy runs while the other half steps, and one half's launch boundary lies under the other half's observation stream.
Compared with one engine over the whole batch, and with the K-steps-per-launch rollout for a policy that needs no
observations (the engine's own uniform random policy).

    python examples/double_buffered_sampling.py [--envs 8192] [--agents 64] [--size 64] [--steps 300]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pogema_amd import GridConfig, PipelinedVecPogema, VecPogema  # noqa: E402


def policy(obs):
    """A stand-in for a network: a cheap, deterministic function of the observation (runs on the current stream).  It
    looks at the 3x3 neighbourhood only, so that the engine and not this function is what the timings below show."""
    c = obs.shape[-1] // 2
    near = obs[:, :, :, c - 1:c + 2, c - 1:c + 2]
    return ((near * torch.arange(1, 28, device=obs.device, dtype=obs.dtype).view(3, 3, 3)).sum(dim=(2, 3, 4)).to(torch.int64) % 5).to(torch.int8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    gc = GridConfig(size=args.size, num_agents=args.agents, obs_radius=5, density=0.3, seed=0, collision_system="soft",
                    max_episode_steps=64)
    n = args.envs * args.agents

    one = VecPogema(gc, batch=args.envs, auto_reset=True, reuse_buffers=True)
    obs, _ = one.reset(seed=0)
    for _ in range(10):
        obs, *_ = one.step(policy(obs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        obs, *_ = one.step(policy(obs))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"one engine            : {dt * 1e6:7.1f} us per step incl. policy = {n / dt / 1e9:.2f} G agent-steps/s")
    one.close()

    pipe = PipelinedVecPogema(gc, batch=args.envs, parts=2, auto_reset=True, reuse_buffers=True)
    obs = [o for o, _ in pipe.reset(seed=0)]

    def sweep():
        for i in range(pipe.parts):
            with pipe.stream(i):
                obs[i], *_ = pipe.step_part(i, policy(obs[i]))

    for _ in range(10):
        sweep()
    pipe.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep()
    pipe.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"two pipelined engines : {dt * 1e6:7.1f} us per step incl. policy = {n / dt / 1e9:.2f} G agent-steps/s")
    pipe.close()

    env = VecPogema(gc, batch=args.envs, auto_reset=True)
    env.reset(seed=0)
    env.rollout(steps=8, obs_slots=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = env.rollout(steps=args.steps, policy_seed=1, obs_slots=2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"rollout, random policy: {dt * 1e6:7.1f} us per step (one launch of {args.steps} steps, ring of 2 observations); "
          f"mean reward {float(out['rewards'].mean()):.4f}")
    env.close()


if __name__ == "__main__":
    main()
