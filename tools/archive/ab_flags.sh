#!/bin/bash
# Diagnostic: same-box A/B of PGX_FLAGS tuning bits over workloads.  usage: tools/ab_flags.sh "0 8" "cfg2 cfg3"
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-32s kernel_ms=%.4f value=%.3e frac=%.3f' % ('$label', r['kernel_ms'], d['value'], r['frac']))"; }
for rep in 1 2; do for wl in ${2:-cfg2}; do for f in ${1:-0 8}; do run "$wl flags=$f" PGX_FLAGS=$f -- --workload $wl --steps 300; done; done; done
