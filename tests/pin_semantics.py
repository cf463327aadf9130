#!/usr/bin/env python3
"""Which position of every semantics switch do reference fixtures demand?  (test infrastructure.)

    python tests/pin_semantics.py DIR          DIR holds reference_*.npz (+ reference_probes.json) of tools/gen_golden.py

Brute force: every combination of the step-semantics switches (soft_vertex x soft_occupancy x coop_reward x bad_action =
2^4) is run through the literal Python oracle over every fixture -- positions, flags, rewards, observations, the occupancy
array and the final metrics, exactly the comparison of tests/test_golden_reference.py -- and the combinations under which
ALL fixtures pass are reported (`--write-pin FILE`: and written as the product's pinned defaults, pogema_amd/semantics.py), per switch: `determined` when every passing combination agrees on it, `free` when the
fixtures cannot tell (e.g. bad_action: rollouts only contain valid actions -- reference_probes.json decides that one).
Prints one JSON object; exit code 0 iff at least one combination passes.  The product's defaults
(pogema_amd.Semantics()) are flagged when they are not among the passing combinations: that is the flip to make.
"""
import glob
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv):
    pin_out = None
    if "--write-pin" in argv:  # write the switch positions the fixtures demand as the product's pinned defaults
        i = argv.index("--write-pin")
        pin_out = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    if len(argv) != 2:
        sys.exit(__doc__)
    os.environ["PGX_GOLDEN_DIR"] = os.path.abspath(argv[1])
    import test_golden_reference as tgr
    from pogema_amd.semantics import BAD_ACTION, COOP_REWARD, SOFT_OCCUPANCY, SOFT_VERTEX, Semantics
    from util import oracle_rollout
    fixtures = [p for p in sorted(glob.glob(os.path.join(os.environ["PGX_GOLDEN_DIR"], "reference_*.npz")))
                if not os.path.basename(p).startswith("reference_grid_")]  # (grid-layer fixtures carry no envs.py semantics)
    if not fixtures:
        sys.exit(f"no reference_*.npz under {argv[1]}")
    probe_file = os.path.join(os.environ["PGX_GOLDEN_DIR"], "reference_probes.json")
    probes = json.load(open(probe_file)) if os.path.exists(probe_file) else {}
    standin = bool(probes.get("standin"))
    if pin_out and standin and os.path.abspath(pin_out).startswith(os.path.join(ROOT, "pogema_amd") + os.sep):
        # fixtures generated from the repo's stand-in package are the builder's own oracle talking to itself;
        # they must never become the product's process-wide defaults (a rehearsal may write its pin anywhere else)
        sys.exit(f"refusing --write-pin {pin_out}: the fixtures under {argv[1]} come from the STAND-IN pogema package "
                 f"(reference_probes.json: standin = true); a stand-in pin may only be written outside pogema_amd/")
    switches = {"soft_vertex": SOFT_VERTEX, "soft_occupancy": SOFT_OCCUPANCY, "coop_reward": COOP_REWARD, "bad_action": BAD_ACTION}
    names = list(switches)
    passing, first_failure = [], {}
    for combo in itertools.product(*switches.values()):
        sem = Semantics(**dict(zip(names, combo)))

        def run(*a, **kw):
            return oracle_rollout(*a, semantics=sem, **kw)

        ok = True
        for path in fixtures:
            try:
                tgr.compare_with_fixture(run, path)
            except (AssertionError, IndexError) as exc:
                ok = False
                msg = next((ln.strip() for ln in str(exc).splitlines() if ln.strip()), type(exc).__name__)
                first_failure[",".join(combo)] = f"{os.path.basename(path)}: {msg[:160]}"
                break
        if ok:
            passing.append(dict(zip(names, combo)))
    verdict = {}
    for n in names:
        seen = sorted({c[n] for c in passing})
        verdict[n] = {"determined": seen[0]} if len(seen) == 1 else {"free": seen} if seen else {"no combination passes": True}
    if probes:
        ba = str(probes.get("bad_action", ""))
        if "free" in verdict.get("bad_action", {}) and ba:
            verdict["bad_action"] = {"determined_by_probe": "flag" if ba.startswith("raises IndexError") else "noop" if ba == "noop" else ba}
    default = {n: getattr(Semantics(), n) for n in names}
    pin = {n: (v.get("determined") or v.get("determined_by_probe")) for n, v in verdict.items()
           if (v.get("determined") or v.get("determined_by_probe")) in switches[n]}
    if pin_out and passing:
        with open(pin_out, "w") as f:
            json.dump({"switches": pin, "fixtures": len(fixtures), "source": os.environ["PGX_GOLDEN_DIR"],
                       "package": probes.get("package", "pogema"), "package_version": probes.get("pogema_version") or probes.get("version"),
                       "standin": standin,
                       "differs_from_recalled_defaults": {n: v for n, v in pin.items() if v != default[n]},
                       "note": "written by tools/pin_reference.sh: the positions of the semantics switches that reference "
                               "fixtures demand; pogema_amd.Semantics.from_env() uses them as the process-wide defaults"}, f, indent=1)
    report = {"pin": pin, "pin_file": pin_out if (pin_out and passing) else None,"fixtures": len(fixtures), "combinations_tried": 2 ** len(names), "passing": passing, "per_switch": verdict,
              "product_default": default, "product_default_passes": default in passing,
              "probes": probes, "first_failure_of_failing_combinations": first_failure}
    print(json.dumps(report, indent=1))
    return 0 if passing else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
