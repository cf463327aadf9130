#!/bin/bash
# Diagnostic: environments-per-wave sweep at large batches (A <= 32 geometries).
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-42s kernel_ms=%.4f value=%.3e frac=%.3f' % ('$label', r['kernel_ms'], d['value'], r['frac']))"; }
for b in 16384 65536; do
for e in 1 2 4; do run "cfg3 batch=$b epw=$e" PGX_EPW=$e -- --workload cfg3 --batch $b --steps 100; done
for e in 1 2 4 8; do run "cfg1 batch=$b epw=$e" PGX_EPW=$e -- --workload cfg1 --batch $b --steps 100; done
done
