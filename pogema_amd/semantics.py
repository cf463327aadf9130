"""Switches for the semantics of the reference that the builder recalls with LOW confidence.

Every DEFAULT is the recalled literal behaviour of upstream, quirks included (bit-exactness targets what upstream does,
not what would be tidier); the other value is the plausible alternative.

The reference source is not mounted (/root/reference/README.md:3,5 -- "code is hosted elsewhere"), so a few details
of upstream `pogema/envs.py` are recollections that the real package may contradict (docs/SPEC.md open questions
Q1, Q2, Q4, Q7).  Each is a configuration switch -- implemented in the step kernel, in both oracles and in the parity
matrix -- so that pinning against the real package is a flip here, not a kernel edit:

    soft_vertex   'lowest_index' (default): collision_system='soft', several movers claim one free or vacated cell:
                                 the lowest agent index moves, the others stay (literal `used_cells[cell].remove(i)`
                                 + reverse-index loop + recursive `_revert_action`, as recalled);
                  'all_stay'   : every claimant of a contested cell stays (the textbook MAPF vertex-conflict rule,
                                 SURVEY.md A5's one-line "net semantics").
    soft_occupancy 'index_order' (default): the literal `move_without_checks` loop as recalled (Q2) -- clear the old cell,
                                 set the new one, agent by agent in index order: an agent that enters the cell a
                                 HIGHER-index agent is leaving gets its new cell cleared again by that agent's turn; it
                                 stands there but is missing from the occupancy array (`Grid.positions`: the `agents`
                                 plane of every observation, `get_state(occupancy=True)`) until its turn in a LATER step
                                 re-sets it -- persistent, as upstream's array is: `observe()`, snapshots and
                                 `step(compute_obs=False)` + `observe()` all show the same array as the step itself;
                  'exact'      : after a `soft` step the occupancy array is exactly the set of visible agents' cells
                                 (the invariant every other collision system keeps).
    coop_reward   'all_solved' (default): on_target='nothing' pays 1.0 to every agent iff ALL agents stand on their
                                 goals (`is_task_solved`);
                  'per_agent'  : 1.0 to each agent standing on its own goal in this step.
    bad_action    'noop' (default): an action outside 0..4 does nothing;
                  'flag'       : the device still treats it as a noop but counts it, and `step()` raises the
                                 reference's IndexError (`MOVES[action]`) -- costs one host sync per step.

    lifelong_rng  'build' (default): the lifelong (on_target='restart') target draw uses the build's counter-based stream;
                  'numpy'      : per-agent numpy generators set up as upstream `PogemaLifeLong._initialize_grid` does
                                 (recalled, conf. medium): main = default_rng(seed_env); seeds = main.integers(2^31-1,
                                 size=A); generator[a] = default_rng(seeds[a]); target = component[generator[a]
                                 .integers(0, len(component))]; re-created at every reset.  seed_env = GridConfig.seed +
                                 global env index.  The numpy arithmetic is bit-exact (pogema_amd/nprng.py); the ORDER of
                                 a component's cells stays build-defined (row-major).

    generator_rng 'build' (default): random instances come from the build's counter-based generator (connectivity-checked
                                 obstacles, rejection-sampled starts/targets; docs/SPEC.md);
                  'numpy'      : instances are drawn as upstream draws them (pogema/generator.py, recalled, conf. medium):
                                 obstacles = default_rng(seed_env).binomial(1, density, (H, W)); free cells shuffled by a
                                 fresh default_rng(seed_env); starts/targets paired along each component in that order
                                 (`placing`).  One GPU thread per env (pgx_np_generate); OverflowError when an env
                                 cannot hold `num_agents` pairs.  A host-only switch: the step kernel never sees it.

`PGX_SEMANTICS="soft_vertex=all_stay,coop_reward=per_agent"` overrides the defaults process-wide (a whole test run
against fixtures from the real package).

PINNED DEFAULTS.  The built-in defaults above are recollections.  Once `tools/pin_reference.sh` has run against the real
package, the positions its fixtures DEMAND are written to `pogema_amd/pinned_semantics.json` (PGX_PINNED_SEMANTICS_FILE
names another file), and from then on they ARE the process-wide defaults (`Semantics.from_env()`, i.e. every VecPogema
built without an explicit `semantics=`): a recollection that turns out wrong is corrected by data, with no edit here
(the `soft_occupancy` default in particular is only as good as the memory it rests on).  Order of precedence:
explicit `semantics=` argument > PGX_SEMANTICS > pinned file > built-in recalled literals.  `pinned_source()` says which
file, if any, is in force; the C-ABI's numbering (0 = recalled literal) is unaffected.
"""
from __future__ import annotations

import os
import json
from dataclasses import dataclass

LIFELONG_RNG = ("build", "numpy")
SOFT_VERTEX = ("lowest_index", "all_stay")
SOFT_OCCUPANCY = ("index_order", "exact")
COOP_REWARD = ("all_solved", "per_agent")
BAD_ACTION = ("noop", "flag")
GENERATOR_RNG = ("build", "numpy")


_SWITCHES = {"soft_vertex": SOFT_VERTEX, "soft_occupancy": SOFT_OCCUPANCY, "coop_reward": COOP_REWARD, "bad_action": BAD_ACTION,
             "lifelong_rng": LIFELONG_RNG, "generator_rng": GENERATOR_RNG}


def pinned_file() -> str:
    return os.environ.get("PGX_PINNED_SEMANTICS_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "pinned_semantics.json")


_PIN_CACHE = {}  # path -> ((mtime_ns, size), parsed switches)


def pinned_defaults() -> dict:
    """{switch: value} demanded by reference fixtures (written by tools/pin_reference.sh), {} when nothing is pinned.  A
    malformed file is an error, not something to fall through silently: the defaults of a pinned build must not depend
    on whether a JSON file happened to parse.  A pin derived from the repo's STAND-IN package (tests/standin_pogema*: built
    from the builder's own oracle) is refused at the DEFAULT location -- it pins nothing; only a file named
    explicitly through PGX_PINNED_SEMANTICS_FILE may be one (the rehearsal of tests/test_golden_pipeline.py).  Parsed once per
    (path, mtime, size)."""
    path = pinned_file()
    try:
        st = os.stat(path)
    except OSError:
        return {}
    stamp = (st.st_mtime_ns, st.st_size)
    hit = _PIN_CACHE.get(path)
    if hit is not None and hit[0] == stamp:
        return dict(hit[1])
    with open(path) as f:
        data = json.load(f)
    if data.get("standin") and not os.environ.get("PGX_PINNED_SEMANTICS_FILE"):  # (an explicitly named file is a rehearsal's own business)
        raise ValueError(f"{path} was derived from the stand-in `pogema` package (fixtures built from this repo's own oracle): "
                         f"it cannot pin the product's semantics -- delete it or point PGX_PINNED_SEMANTICS_FILE elsewhere")
    out = {}
    for k, v in (data.get("switches") or {}).items():
        if k not in _SWITCHES or v not in _SWITCHES[k]:
            raise ValueError(f"{path}: {k}={v!r} is not a known semantics switch / value")
        out[k] = v
    _PIN_CACHE[path] = (stamp, dict(out))
    return out


def pinned_source():
    """Path of the pinned-defaults file in force, or None (the defaults are the builder's recollections)."""
    return pinned_file() if pinned_defaults() else None


@dataclass(frozen=True)
class Semantics:
    soft_vertex: str = "lowest_index"
    soft_occupancy: str = "index_order"
    coop_reward: str = "all_solved"
    bad_action: str = "noop"
    lifelong_rng: str = "build"
    generator_rng: str = "build"

    def __post_init__(self):
        for name, allowed in (("soft_vertex", SOFT_VERTEX), ("soft_occupancy", SOFT_OCCUPANCY), ("coop_reward", COOP_REWARD), ("bad_action", BAD_ACTION),
                              ("lifelong_rng", LIFELONG_RNG), ("generator_rng", GENERATOR_RNG)):
            if getattr(self, name) not in allowed:
                raise ValueError(f"Semantics.{name} must be one of {allowed}, got {getattr(self, name)!r}")

    @classmethod
    def from_env(cls) -> "Semantics":
        """The process-wide default: built-in recalled literals < pinned file (reference fixtures) < PGX_SEMANTICS."""
        spec = os.environ.get("PGX_SEMANTICS", "").strip()
        kw = pinned_defaults()
        for item in filter(None, (s.strip() for s in spec.split(","))):
            if "=" not in item:
                raise ValueError(f"PGX_SEMANTICS entry {item!r} is not key=value")
            k, v = item.split("=", 1)
            if k not in ("soft_vertex", "soft_occupancy", "coop_reward", "bad_action", "lifelong_rng", "generator_rng"):
                raise ValueError(f"PGX_SEMANTICS: unknown switch {k!r}")
            kw[k] = v
        return cls(**kw)

    def oracle_kwargs(self) -> dict:
        """The same switches under the oracles' parameter names (tests only)."""
        return {"soft_vertex_rule": self.soft_vertex, "soft_occupancy": self.soft_occupancy, "coop_reward": self.coop_reward, "bad_action": self.bad_action,
                "lifelong_rng": self.lifelong_rng}
