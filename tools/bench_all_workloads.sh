#!/bin/bash
# benches of every BASELINE workload with the box fingerprint, the headline under both walk budgets, profiles
O=gpurun_out/${1:-r4f}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --placement-budget all --no-cpu-baseline --no-extras > $O/bench_cfg2_budget_all.json 2> $O/bench_cfg2_budget_all.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_driver_form.json 2> $O/bench_cfg2_driver_form.err
for wl in cfg3 cfg1 cfg4; do python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; done
bash tools/collect_profiles.sh $O/prof cfg1 cfg2 cfg3 cfg4 > $O/collect.log 2>&1
python - <<'PY'
import json, glob, os, sys
O = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("O", "")
PY
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1], "ERR", e); sys.exit(0)
r = d["roofline"]; pl = r.get("placement") or {}
sec = {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d.get("secondary", {}).items() if isinstance(v, dict) and "ms_per_step" in v}
print(f"{sys.argv[1].split('/')[-1]:34s} step {d['ms_per_step']*1e3:7.1f} us kernel {r['kernel_ms']*1e3:7.1f} frac {r['frac']:.3f} default-placement {(r.get('default_placement_kernel_ms') or 0)*1e3:6.1f} spread={pl.get('spread')} cand={pl.get('walk_candidates')} budget={pl.get('budget_gib')} {sec}")
PY
done
tail -6 $O/collect.log
