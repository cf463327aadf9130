"""Double-buffered sampling over one batch: S engines, each on its own HIP stream.

A step launch is one resident round of waves: all of them run their state phase at the start (HBM idle) and drain at the
end together, and consecutive launches of ONE stream cannot overlap.  Two engines over the two halves of the batch,
driven alternately by the usual double-buffered loop

    for i in range(env.parts):                       # Sample Factory's double-buffered sampling, PyMARL's parallel runner
        with env.stream(i):
            actions = policy(obs[i])                 # ... runs while the other half steps
            obs[i], rew, term, trunc, infos = env.step_part(i, actions)

drift out of phase, and the launch boundary of one half lies under the observation stream of the other: configs[3]
43 -> 33 us, configs[4] 500 -> 450 us, configs[2] 128 -> 110..123 us per full batch (profiles/r2/split_streams.txt).
Forcing the halves back into lockstep (a device-side join after every step) loses all of it -- which is why this is a
separate object with per-part calls and not something step() of a single engine could do behind the caller's back.

Part i holds the global envs [i * batch / S, (i + 1) * batch / S): the instances, the lifelong target streams and every
result are those of ONE engine over the whole batch (tests/test_pipeline_gpu.py).
"""
from __future__ import annotations

from typing import Optional

import torch

from .grid_config import GridConfig
from .vec_env import VecPogema


class PipelinedVecPogema:
    def __init__(self, grid_config: Optional[GridConfig] = None, batch: int = 2, device="cuda:0", parts: int = 2,
                 env_index_base: int = 0, **engine_kwargs):
        if parts < 1 or batch % parts != 0:
            raise ValueError(f"batch ({batch}) must be a positive multiple of parts ({parts})")
        self.parts = int(parts)
        self.batch = int(batch)
        self.part_batch = self.batch // self.parts
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.parts)]
        self.engines = []
        for i in range(self.parts):
            with torch.cuda.stream(self.streams[i]):
                self.engines.append(VecPogema(grid_config, batch=self.part_batch, device=device,
                                              env_index_base=env_index_base + i * self.part_batch, **engine_kwargs))
        self.grid_config = self.engines[0].grid_config
        self.num_agents = self.engines[0].num_agents
        self.obs_shape = (self.batch,) + tuple(self.engines[0].obs_shape[1:])

    def stream(self, i: int):
        """Context manager: torch work issued inside runs on part i's stream (policy inference for that half)."""
        return torch.cuda.stream(self.streams[i])

    def part_slice(self, i: int) -> slice:
        return slice(i * self.part_batch, (i + 1) * self.part_batch)

    def reset(self, seed: Optional[int] = None):
        """Resets every part (same instances as one engine over the whole batch); returns a list of (obs, infos)."""
        out = []
        for i, e in enumerate(self.engines):
            with self.stream(i):
                out.append(e.reset(seed=seed))
        return out

    def warm_buffers(self):
        """`VecPogema.warm_buffers` of every part (zone walk and candidate timing outside the sampling loop)."""
        out = []
        for i, e in enumerate(self.engines):
            with self.stream(i):
                out.append(e.warm_buffers())
        return out

    def step_part(self, i: int, actions, **kw):
        """`VecPogema.step` of part i, enqueued on part i's stream; `actions`: [batch / parts, agents].  If the actions
        were produced on another stream, make part i's stream wait for them first (`wait_for`)."""
        with self.stream(i):
            return self.engines[i].step(actions, **kw)

    def step(self, actions, **kw):
        """All parts, one after the other, each on its own stream, WITHOUT joining them: `actions` is a
        [batch, agents] tensor (split along the batch) or a list of per-part tensors; returns the list of per-part
        results.  The caller reads part i's tensors on stream i or after `wait_part(i)` / `synchronize()`."""
        if isinstance(actions, torch.Tensor):
            if actions.shape[0] != self.batch:
                raise ValueError(f"actions must have {self.batch} rows")
            actions = [actions[self.part_slice(i)] for i in range(self.parts)]
        if len(actions) != self.parts:
            raise ValueError(f"need {self.parts} action tensors")
        return [self.step_part(i, a, **kw) for i, a in enumerate(actions)]

    def wait_for(self, i: int, stream=None):
        """Part i's stream waits for everything enqueued so far on `stream` (default: the current stream)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.streams[i].wait_stream(s)

    def wait_part(self, i: int, stream=None):
        """`stream` (default: the current stream) waits for everything enqueued so far on part i's stream."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        s.wait_stream(self.streams[i])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def get_state(self):
        """Per-part state dicts concatenated along the batch axis (synchronises)."""
        states = []
        for i, e in enumerate(self.engines):
            with self.stream(i):
                states.append(e.get_state())
        self.synchronize()
        return {k: torch.cat([s[k] for s in states]) for k in states[0]}

    def close(self):
        self.synchronize()
        for e in self.engines:
            e.close()
