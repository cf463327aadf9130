"""Single-environment, list-per-agent view of the engine: the reference's native calling convention
(upstream `pogema/envs.py` + `pogema/integrations/make_pogema.py`: `pogema_v0(grid_config)`;
SURVEY.md section 8b "what the reference exposes").

    env = pogema_v0(GridConfig(num_agents=2, size=8))
    obs, infos = env.reset()
    obs, rewards, terminated, truncated, infos = env.step([0, 3])

`obs[i]` is a float32 numpy array (3, 2r+1, 2r+1); rewards floats; terminated/truncated bools; infos
dicts with 'is_active'.  Everything is computed by the HIP engine with batch = 1 (this view exists
for API compatibility and tests, not for throughput -- use `VecPogema` for that).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .grid_config import GridConfig
from .vec_env import VecPogema

try:  # gymnasium is optional (not installed in the build image)
    import gymnasium  # type: ignore
    _Box, _Discrete = gymnasium.spaces.Box, gymnasium.spaces.Discrete
except Exception:  # pragma: no cover - exercised only where gymnasium is missing
    class _Box:  # minimal duck-typed stand-ins
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

    class _Discrete:
        def __init__(self, n):
            self.n = n
            self._rng = np.random.default_rng()

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n


class Pogema:
    def __init__(self, grid_config: Optional[GridConfig] = None, device="cuda:0", auto_reset=None, semantics=None):
        self.grid_config = grid_config if grid_config is not None else GridConfig(num_agents=2)
        self._vec = VecPogema(self.grid_config, batch=1, device=device, auto_reset=auto_reset, semantics=semantics)
        full = 2 * self.grid_config.obs_radius + 1
        self.observation_space = _Box(0.0, 1.0, shape=(3, full, full), dtype=np.float32)
        self.action_space = _Discrete(len(self.grid_config.MOVES))
        # `GridConfig.persistent` (upstream PersistentWrapper): one engine snapshot per step, `step_back()` restores
        self._history = [] if self.grid_config.persistent else None

    def get_num_agents(self):
        return self.grid_config.num_agents

    def sample_actions(self):
        return [self.action_space.sample() for _ in range(self.get_num_agents())]

    def _obs_list(self, obs):
        if isinstance(obs, dict):  # POMAPF / MAPF: one dict per agent, as the reference returns
            host = {k: v[0].cpu().numpy() for k, v in obs.items()}
            n = self.get_num_agents()
            per_agent = []
            for i in range(n):
                d = {k: (host[k] if k == "global_obstacles" else host[k][i]) for k in host}
                for k in ("xy", "target_xy", "global_xy", "global_target_xy"):
                    if k in d:
                        d[k] = tuple(int(c) for c in d[k])
                per_agent.append(d)
            return per_agent
        host = obs[0].cpu().numpy()
        return [host[i] for i in range(host.shape[0])]

    def reset(self, seed: Optional[int] = None, return_info: bool = True, options=None):
        if self._history is not None:
            self._history.clear()
        obs, infos = self._vec.reset(seed=seed)
        active = infos["is_active"][0].cpu().numpy()
        out = self._obs_list(obs)
        if return_info:
            return out, [{"is_active": bool(a)} for a in active]
        return out

    def _packed_io(self):
        """One device buffer for every per-step output (+ a pinned host mirror) and a pinned staging buffer for the
        actions: a step of this single-env view then costs one H2D copy, one kernel launch and ONE D2H copy instead
        of six small synchronous copies (135 -> ~45 us per step on an MI355X box)."""
        if getattr(self, "_io", None) is not None:
            return self._io
        import torch
        vec = self._vec
        A, W = vec.num_agents, vec.window
        n_obs = A * 3 * W * W * 4
        off = {"obs": 0, "rewards": n_obs, "metrics": n_obs + 4 * A}
        off["terminated"] = off["metrics"] + 4 * 6
        off["truncated"] = off["terminated"] + A
        off["is_active"] = off["truncated"] + A
        off["episode_done"] = off["is_active"] + A
        total = off["episode_done"] + 1
        dev = torch.zeros(total, dtype=torch.uint8, device=vec.device)
        host = torch.zeros(total, dtype=torch.uint8).pin_memory()
        hnp = host.numpy()
        views = {
            "obs": hnp[:n_obs].view(np.float32).reshape(A, 3, W, W),
            "rewards": hnp[off["rewards"]:off["rewards"] + 4 * A].view(np.float32),
            "metrics": hnp[off["metrics"]:off["metrics"] + 24].view(np.float32),
            "terminated": hnp[off["terminated"]:off["terminated"] + A], "truncated": hnp[off["truncated"]:off["truncated"] + A],
            "is_active": hnp[off["is_active"]:off["is_active"] + A], "episode_done": hnp[off["episode_done"]:],
        }
        act_host = torch.zeros(A, dtype=torch.int64).pin_memory()
        act_dev = torch.zeros(A, dtype=torch.int64, device=vec.device)
        base = dev.data_ptr()
        ptr = {k: base + v for k, v in off.items()}
        from . import _lib
        _lib.check(vec._lib.pgx_set_metrics_buffers(vec._handle, ptr["metrics"], ptr["episode_done"]))
        self._io = dict(dev=dev, host=host, views=views, act_host=act_host, act_np=act_host.numpy(), act_dev=act_dev, ptr=ptr)
        return self._io

    def step_back(self) -> bool:
        """Undo the last step (`PersistentWrapper.step_back`); needs GridConfig(persistent=True)."""
        if self._history is None:
            raise RuntimeError("step_back needs GridConfig(persistent=True)")
        if not self._history:
            return False
        self._vec.load_state(self._history.pop())
        return True

    def _refuse_bad_actions(self, action):
        """Semantics(bad_action='flag'): the reference indexes MOVES[action] for every ACTIVE agent before anything
        moves, so an out-of-range action raises IndexError with the state untouched.  The actions of this API are host
        values: check them here, BEFORE the engine steps (the is_active flags are fetched only when something is out
        of range).  Python's negative wrap-around (MOVES[-1]) is not reproduced: negative actions are refused."""
        acts = np.asarray(action).reshape(-1)
        out_of_range = (acts < 0) | (acts >= len(self.grid_config.MOVES))
        if out_of_range.any():
            active = self._vec.get_state()["is_active"][0].cpu().numpy().astype(bool)
            bad = int((out_of_range & active).sum())
            if bad:
                raise IndexError(f"{bad} action(s) of active agents were outside 0..{len(self.grid_config.MOVES) - 1}")

    def step(self, action):
        assert len(action) == self.get_num_agents()
        if self._vec.semantics.bad_action == "flag":
            self._refuse_bad_actions(action)
        if self._history is not None:
            self._history.append(self._vec.save_state())
        import torch
        from . import _lib
        vec = self._vec
        if vec.observation_type != "default":  # dict observations: the general (multi-copy) route
            return self._step_general(action)
        io = self._packed_io()
        io["act_np"][:] = action
        io["act_dev"].copy_(io["act_host"], non_blocking=True)
        p = io["ptr"]
        _lib.check(vec._lib.pgx_step(vec._handle, io["act_dev"].data_ptr(), 2, p["obs"], p["rewards"], p["terminated"],
                                     p["truncated"], p["is_active"], vec._stream()))
        io["host"].copy_(io["dev"], non_blocking=True)
        torch.cuda.current_stream(vec.device).synchronize()
        if vec.semantics.bad_action == "flag":  # the reference's IndexError on MOVES[action]
            bad = int(vec._lib.pgx_bad_action_count(vec._handle, vec._stream()))
            if bad < 0:
                _lib.check(bad)
            if bad:  # unreachable after _refuse_bad_actions; kept as the engine's own verdict
                raise IndexError(f"{bad} action(s) of active agents were outside 0..{len(self.grid_config.MOVES) - 1}")
        v = io["views"]
        obs = v["obs"].copy()
        info_list = [{"is_active": bool(a)} for a in v["is_active"]]
        if v["episode_done"][0]:  # metric wrappers: infos[0]['metrics'] on the step that ends the episode
            info_list[0]["metrics"] = self._metrics_dict(v["metrics"])
        return ([obs[i] for i in range(obs.shape[0])], [float(x) for x in v["rewards"]],
                [bool(x) for x in v["terminated"]], [bool(x) for x in v["truncated"]], info_list)

    def _metrics_dict(self, values):
        from ._lib import METRIC_NAMES
        metrics = {k: float(x) for k, x in zip(METRIC_NAMES, values)}
        if self.grid_config.on_target == "restart":
            return {"avg_throughput": metrics["avg_throughput"]}
        metrics.pop("avg_throughput")
        return metrics

    def _step_general(self, action):
        obs, rewards, terminated, truncated, infos = self._vec.step(np.asarray(action, dtype=np.int64)[None])
        info_list = [{"is_active": bool(v)} for v in infos["is_active"][0].cpu().numpy()]
        if bool(infos["episode_done"][0]):
            info_list[0]["metrics"] = self._metrics_dict(infos["metrics"][0].cpu().numpy())
        return (self._obs_list(obs), [float(v) for v in rewards[0].cpu().numpy()],
                [bool(v) for v in terminated[0].cpu().numpy()], [bool(v) for v in truncated[0].cpu().numpy()],
                info_list)

    # ---- state accessors in the style of `Grid.get_agents_xy` etc. (unpadded coordinates) ----------
    def get_agents_xy(self):
        return [tuple(int(c) for c in p) for p in self._vec.get_state()["agents_xy"][0].cpu().numpy()]

    def get_targets_xy(self):
        return [tuple(int(c) for c in p) for p in self._vec.get_state()["targets_xy"][0].cpu().numpy()]

    def get_obstacles(self):
        """Unpadded obstacle map (`Grid.get_obstacles(ignore_borders=True)`), int array [H, W]."""
        return self._vec._initial[0][0].cpu().numpy().astype(np.int64)

    def get_agents_xy_relative(self):
        """Agent cells relative to their start cells (`Grid.get_agents_xy_relative`)."""
        start = self._vec._initial[1][0].cpu().numpy()
        return [(int(x - sx), int(y - sy)) for (x, y), (sx, sy) in zip(self.get_agents_xy(), start)]

    def get_targets_xy_relative(self):
        start = self._vec._initial[1][0].cpu().numpy()
        return [(int(x - sx), int(y - sy)) for (x, y), (sx, sy) in zip(self.get_targets_xy(), start)]

    def get_state(self):
        """`Grid.get_state`-style export: obstacles, agents, targets, active flags (unpadded coordinates)."""
        st = self._vec.get_state()
        return {"obstacles": self.get_obstacles(), "agents_xy": self.get_agents_xy(), "targets_xy": self.get_targets_xy(),
                "is_active": [bool(v) for v in st["is_active"][0].cpu().numpy()], "elapsed": int(st["elapsed"][0])}

    def render(self, mode: str = "ansi"):
        """Text rendering of the current state (upstream `pogema/utils.py: render_grid`, console view): '#' obstacle,
        '.' free, lowercase letter = agent, the same letter in uppercase = its target ('*' agent standing on its own
        target; hidden/finished agents are not drawn).  Returns the string (and prints it when mode == 'human')."""
        obstacles = self.get_obstacles()
        st = self.get_state()
        rows = [["#" if v else "." for v in row] for row in obstacles]
        names = "abcdefghijklmnopqrstuvwxyz"
        for i, (tx, ty) in enumerate(st["targets_xy"]):
            rows[tx][ty] = names[i % 26].upper()
        for i, ((x, y), active) in enumerate(zip(st["agents_xy"], st["is_active"])):
            if active:
                rows[x][y] = "*" if (x, y) == tuple(st["targets_xy"][i]) else names[i % 26]
        text = "\n".join("".join(r) for r in rows)
        if mode == "human":
            print(text)
        return text

    def close(self):
        self._vec.close()


class PogemaParallel:
    """PettingZoo-parallel-style dict view (agent names 'player_i'); thin adapter over `Pogema`."""

    def __init__(self, grid_config: Optional[GridConfig] = None, device="cuda:0"):
        self._env = Pogema(grid_config, device=device)
        self.possible_agents = [f"player_{i}" for i in range(self._env.get_num_agents())]
        self.agents = list(self.possible_agents)

    def observation_space(self, agent):
        return self._env.observation_space

    def action_space(self, agent):
        return self._env.action_space

    def reset(self, seed=None, options=None):
        obs, infos = self._env.reset(seed=seed)
        self.agents = list(self.possible_agents)
        return dict(zip(self.possible_agents, obs)), dict(zip(self.possible_agents, infos))

    def step(self, actions: dict):
        acts = [int(actions.get(name, 0)) for name in self.possible_agents]
        obs, rew, term, trunc, infos = self._env.step(acts)
        names = self.possible_agents
        self.agents = [n for n, t, tr in zip(names, term, trunc) if not (t or tr)]
        return (dict(zip(names, obs)), dict(zip(names, rew)), dict(zip(names, term)), dict(zip(names, trunc)),
                dict(zip(names, infos)))


class PogemaSampleFactory(Pogema):
    """`integration='SampleFactory'` view (upstream `pogema/integrations/sample_factory.py`: AutoResetWrapper over
    IsMultiAgentWrapper over MetricsForwardingWrapper): multi-agent lists, auto-reset inside `step()` (done by the
    engine), `is_multiagent` / `num_agents` attributes, episode metrics copied to `infos[i]['episode_extra_stats']`."""

    is_multiagent = True

    def __init__(self, grid_config: Optional[GridConfig] = None, device="cuda:0"):
        super().__init__(grid_config, device=device, auto_reset=True)

    @property
    def num_agents(self):
        return self.get_num_agents()

    def step(self, action):
        obs, rewards, terminated, truncated, infos = super().step(action)
        for info in infos:
            if "metrics" in info:
                info["episode_extra_stats"] = dict(info["metrics"])
        return obs, rewards, terminated, truncated, infos


class PogemaSingleAgent:
    """`integration='gymnasium'` view for `num_agents == 1` (upstream `SingleAgentWrapper`): plain gymnasium
    signature with scalar reward / flags and one observation array."""

    def __init__(self, grid_config: Optional[GridConfig] = None, device="cuda:0"):
        self._env = Pogema(grid_config, device=device)
        if self._env.get_num_agents() != 1:
            raise ValueError("integration='gymnasium' is the single-agent view: num_agents must be 1")
        self.observation_space, self.action_space = self._env.observation_space, self._env.action_space
        self.grid_config = self._env.grid_config

    def reset(self, seed: Optional[int] = None, options=None):
        obs, infos = self._env.reset(seed=seed)
        return obs[0], infos[0]

    def step(self, action):
        obs, rew, term, trunc, infos = self._env.step([int(action)])
        return obs[0], rew[0], term[0], trunc[0], infos[0]

    def close(self):
        self._env.close()


def pogema_v0(grid_config: Optional[GridConfig] = None, device="cuda:0", semantics=None):
    """Factory with the reference's name: dispatches on `GridConfig.integration`.  `semantics`
    (pogema_amd.Semantics) selects the variants of the low-confidence recollections; default: PGX_SEMANTICS or the
    recalled behaviour."""
    gc = grid_config if grid_config is not None else GridConfig(num_agents=2)
    if gc.integration is None:
        return Pogema(gc, device=device, semantics=semantics)
    if gc.integration == "gymnasium":
        return PogemaSingleAgent(gc, device=device)
    if gc.integration == "PettingZoo":
        return PogemaParallel(gc, device=device)
    if gc.integration == "SampleFactory":
        return PogemaSampleFactory(gc, device=device)
    raise NotImplementedError(f"integration={gc.integration!r} is outside the hot-path scope of this build")
