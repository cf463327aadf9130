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

`obs_parents` (round 6): FULL-BATCH observation tensors the caller already owns -- typically the output sets of a
VecPogema over the whole batch, placed and timed once.  The parts then write their rows of those tensors in turn
(step k of every part goes to parent k % len(parents)) instead of each part placing buffers of its own with a walk of its
own: the batch's observations land in one contiguous [batch, agents, 3, W, W] tensor, and the placement verdict of the
parent engine is the pipeline's (bench.py's `pipelined` figure no longer rolls its own dice).
"""
from __future__ import annotations

from typing import Optional

import torch

from .grid_config import GridConfig
from .vec_env import VecPogema


class PipelinedVecPogema:
    def __init__(self, grid_config: Optional[GridConfig] = None, batch: int = 2, device="cuda:0", parts: int = 2,
                 env_index_base: int = 0, obs_parents=None, **engine_kwargs):
        if parts < 1 or batch % parts != 0:
            raise ValueError(f"batch ({batch}) must be a positive multiple of parts ({parts})")
        if obs_parents is not None:
            engine_kwargs = dict(engine_kwargs, reuse_buffers=False)  # every step gets out=(rows of a parent, ...)
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
        self._parents, self._sets, self._turn = None, None, [0] * self.parts
        self._shared_bufs, self._shared_placement = None, None  # reuse_buffers=True: one placement for all parts (_shared_walk)
        if obs_parents is not None:
            obs_parents = list(obs_parents)
            e0 = self.engines[0]
            for t in obs_parents:
                if (tuple(t.shape) != self.obs_shape or t.dtype != e0.obs_dtype or not t.is_contiguous()
                        or t.device != e0.device):
                    raise ValueError(f"obs_parents must be contiguous {e0.obs_dtype} tensors of shape {self.obs_shape} on {e0.device}")
            if not obs_parents:
                raise ValueError("obs_parents is empty")
            self._parents = obs_parents
            B, A, dev = self.batch, self.num_agents, self.device
            # per parent: the small per-step outputs of the whole batch, written by the parts row-wise as well
            self._sets = [(t, torch.empty((B, A), dtype=torch.float32, device=dev), torch.empty((B, A), dtype=torch.bool, device=dev),
                           torch.empty((B, A), dtype=torch.bool, device=dev), torch.empty((B, A), dtype=torch.bool, device=dev))
                          for t in obs_parents]
            # (row views made once: five slices per step and part would be ~10 us of host time in the loop)
            self._rows = [[tuple(t[self.part_slice(i)] for t in st) for i in range(self.parts)] for st in self._sets]

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
        """`VecPogema.warm_buffers` of every part (zone walk and candidate timing outside the sampling loop).  With
        `obs_parents` there is nothing to place: only the parts' XCD shares are tuned on their rows of the parents."""
        out = []
        for i, e in enumerate(self.engines):
            with self.stream(i):
                if self._parents is not None:
                    sl = self.part_slice(i)
                    info = {}
                    if e.batch >= 2048 and e._has_state():
                        info = e.tune_xcd_shares(self._parents[0][sl], self._parents[-1][sl] if len(self._parents) > 1 else None)
                    out.append(dict(info, method="rows of the caller's full-batch tensors (obs_parents)"))
                elif self._shared_walk(i):
                    out.append(e.placement or {})
                else:
                    out.append(e.warm_buffers())
        return out

    def _shared_walk(self, i: int) -> bool:
        """reuse_buffers=True: ONE placement for all parts (round 6) -- part 0 picks 2 x parts observation buffers of the
        part shape in a single walk (candidates timed, the best kept, as for a lone engine) and every part adopts its
        pair, instead of `parts` engines each holding half of the free memory for a walk of their own and each rolling
        their own verdict.  True when part i got (or already had) its buffers this way."""
        e0 = self.engines[0]
        if self.parts < 2 or not e0.reuse_buffers or e0.single_buffer or not e0._has_state():
            return False
        if any(e._bufs is not None for e in self.engines) and self._shared_bufs is None:
            return False  # somebody has placed buffers already (a step before warm_buffers)
        if self._shared_bufs is None:
            with self.stream(0):
                self._shared_bufs = e0._pick_obs_buffers(2 * self.parts)
                self._shared_placement = dict(e0.placement or {}, shared_by_parts=self.parts)
        if self.engines[i]._bufs is None:
            self.engines[i].adopt_obs_buffers(self._shared_bufs[2 * i:2 * i + 2], self._shared_placement)
            self.engines[i]._zone_ptrs = set(e0._zone_ptrs)
        return True

    def parent_outputs(self, k: int):
        """(obs, rewards, terminated, truncated, is_active) of parent k over the WHOLE batch (obs_parents only); rows of
        part i are valid on stream i / after `wait_part(i)`."""
        return self._sets[k]

    def step_part(self, i: int, actions, **kw):
        """`VecPogema.step` of part i, enqueued on part i's stream; `actions`: [batch / parts, agents].  If the actions
        were produced on another stream, make part i's stream wait for them first (`wait_for`)."""
        with self.stream(i):
            if self._sets is not None and "out" not in kw:
                k = self._turn[i]
                self._turn[i] = (k + 1) % len(self._sets)
                kw = dict(kw, out=self._rows[k][i])
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
