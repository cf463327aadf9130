#!/bin/bash
# Fuzz campaign over the round-6 kernel variants (runs ON THE GPU BOX): tests/test_fuzz_gpu.py -- random step configurations
# incl. semantics switches and bad actions, every case also as one rollout launch -- with the resolver / streamer pair forced
# (PGX_ROLL_PC=1), the large-map layout forced (PGX_BIG=1), both, and the re-staging rollout layout (PGX_ROLL_RESIDENT=0).
# usage: tools/fuzz_variants.sh <first seed> <last seed> <step> [outfile]
first=${1:-40000}; last=${2:-41000}; step=${3:-100}; out=${4:-gpurun_out/fuzz_variants.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
fail=0
for v in "PGX_ROLL_PC=1" "PGX_BIG=1" "PGX_ROLL_PC=1 PGX_WAVES=1" "PGX_ROLL_RESIDENT=0" "PGX_BIG=1 PGX_WAVES=8"; do
  for s in $(seq $first $step $last); do
    r=$(env $v PGX_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -1)
    echo "$v PGX_FUZZ_SEED=$s: $r" | tee -a "$out"
    case "$r" in *failed*|*error*) fail=1;; esac
  done
done
echo "variants campaign $first..$last step $step: $([ $fail = 0 ] && echo CLEAN || echo FAILURES)" | tee -a "$out"
exit $fail
