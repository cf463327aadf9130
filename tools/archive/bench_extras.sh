#!/bin/bash
# bench lines with secondary figures for the three GPU configurations (round-2 and round-3 evidence)
mkdir -p gpurun_out
for wl in cfg2 cfg3 cfg4 cfg1; do
  steps=2000; [ $wl = cfg4 ] && steps=500
  PGX_DEBUG=${PGX_DEBUG:-} python bench.py --workload $wl --steps $steps --no-cpu-baseline > gpurun_out/bench_x_$wl.json 2> gpurun_out/bench_x_$wl.err
  python - "$wl" <<'PY'
import json, sys
wl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/bench_x_{wl}.json").read().strip().splitlines()[-1])
except Exception as e:
    print(wl, "FAILED", e); print(open(f"gpurun_out/bench_x_{wl}.err").read()[-1500:]); sys.exit(0)
r = d["roofline"]
print(f"{wl}: step {d['ms_per_step']*1e3:.1f} us frac {r['frac']:.3f} (default placement {(r['default_placement_kernel_ms'] or 0)*1e3:.1f} us) | " +
      " | ".join(f"{k} {v['ms_per_step']*1e3:.1f} us frac {v['frac']:.3f}" for k, v in d.get("secondary", {}).items()))
print("   ", d["config"]["obs_buffers"][:200])
PY
done
