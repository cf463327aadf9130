#!/bin/bash
# Diagnostic: same-box A/B of two engine builds / tuning knobs over the BASELINE workloads.
# usage: tools/ab_bench.sh  (prints workload, variant, kernel_ms, value, frac)
run() { # label, env..., -- args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-42s kernel_ms=%.4f ms_per_step=%.4f value=%.3e frac=%.3f' % ('$label', r['kernel_ms'], d['ms_per_step'], d['value'], r['frac']))"
}
OLD=$PWD/pogema_amd/libpogema_amd_old.so
for wl in cfg2 cfg4 cfg3 cfg1; do
  [ -f $OLD ] && run "$wl old" PGX_LIB=$OLD -- --workload $wl --steps 200
  run "$wl new" X=1 -- --workload $wl --steps 200
done
for e in 1 2 4; do run "cfg3 new epw=$e" PGX_EPW=$e -- --workload cfg3 --steps 200; done
for e in 1 2 4 8; do run "cfg1 new epw=$e" PGX_EPW=$e -- --workload cfg1 --steps 200; done
