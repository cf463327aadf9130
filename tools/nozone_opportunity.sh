#!/bin/bash
# Runs ON THE GPU BOX at the start of a gpurun call: a short headline run; if -- and only if -- this
# lease's full-budget walk found no second zone, collect the tier-tagged profile of configs[2] (kernel stats, FETCH/WRITE
# passes, write-stall counters) into gpurun_out/<dir>/nozone.  Nobody goes looking for such a lease: on a zone lease this
# costs ~15 s.   usage: tools/nozone_opportunity.sh <dir under gpurun_out>
O=gpurun_out/${1:-r5}/nozone_check; mkdir -p $O
python bench.py --no-extras --no-cpu-baseline --no-default-placement --steps 400 --windows 2 > $O/bench.json 2> $O/bench.err
TIER=$(python - "$O/bench.json" <<'PY'
import json, sys
try:
    pl = json.loads(open(sys.argv[1]).readline())["roofline"]["placement"]
    print("nozone" if (not pl["spread"] and (pl["walk_candidates"] or 0) >= 8) else "zone")
except Exception:
    print("unknown")
PY
)
echo "nozone_opportunity: this lease is '$TIER'"
if [ "$TIER" = nozone ]; then
  bash tools/collect_profiles.sh gpurun_out/${1:-r5}/nozone cfg2 > $O/collect.log 2>&1
  tail -3 $O/collect.log
fi
