#!/bin/bash
# Extended fuzz campaign on the GPU box: tests/test_fuzz_gpu.py (random step configurations incl. semantics switches and
# bad actions, every case also as one rollout launch; random device resets) under a range of PGX_FUZZ_SEED values.
# usage: tools/fuzz_campaign.sh <first> <last> <step> [outfile]
first=${1:-6000}; last=${2:-9000}; step=${3:-100}; out=${4:-gpurun_out/fuzz_campaign.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
fail=0
for s in $(seq $first $step $last); do
  r=$(PGX_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -1)
  echo "PGX_FUZZ_SEED=$s: $r" | tee -a "$out"
  case "$r" in *failed*|*error*) fail=1;; esac
done
echo "campaign $first..$last step $step: $([ $fail = 0 ] && echo CLEAN || echo FAILURES)" | tee -a "$out"
exit $fail
