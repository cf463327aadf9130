#!/usr/bin/env python3
"""The observation format a mixed-precision policy wants: bfloat16 planes straight from the step kernel.

A small convolutional policy runs in bfloat16; with `obs_dtype=torch.bfloat16` the engine writes its input in that format
(0.0 and 1.0 are exact), so the environment step moves half the HBM bytes of the float32 mode and the policy reads half --
and no cast pass sits in between (uint8 + `.to(torch.bfloat16)` would move as many bytes as float32 did).

    python examples/bf16_policy_loop.py [--envs 4096] [--agents 32] [--size 32] [--steps 200]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from pogema_amd import GridConfig, VecPogema  # noqa: E402


class TinyPolicy(torch.nn.Module):
    def __init__(self, window):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(),
                                       torch.nn.Linear(16 * window * window, 5))

    def forward(self, obs):                         # obs: [envs, agents, 3, W, W] in the network's own dtype
        b, a = obs.shape[:2]
        return self.net(obs.flatten(0, 1)).view(b, a, 5)


def run(dtype, args):
    gc = GridConfig(size=args.size, num_agents=args.agents, obs_radius=5, density=0.3, seed=0, collision_system="soft",
                    max_episode_steps=64)
    env = VecPogema(gc, batch=args.envs, auto_reset=True, obs_dtype=dtype)
    policy = TinyPolicy(env.window).to(env.device).to(torch.bfloat16)
    obs, _ = env.reset(seed=0)

    def act(o):
        with torch.no_grad():
            return policy(o if o.dtype == torch.bfloat16 else o.to(torch.bfloat16)).argmax(-1).to(torch.int8)

    for _ in range(10):
        obs, *_ = env.step(act(obs))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        obs, rewards, terminated, truncated, infos = env.step(act(obs))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.close()
    return args.steps * args.envs * args.agents / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=32)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    for name, dtype in (("float32 observations + cast", torch.float32), ("uint8 observations + cast", torch.uint8),
                        ("bfloat16 observations", torch.bfloat16)):
        print(f"{name:30s} {run(dtype, args):.3e} agent-steps/s (policy included)")


if __name__ == "__main__":
    main()
