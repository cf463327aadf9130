#!/bin/bash
# Runs ON THE GPU BOX (under gpurun): rocprofv3 evidence for pgx::rollout_kernel (round 6).
#   --kernel-trace --stats of tools/rollout_target.py           -> average duration of one K-step launch
#   separate --pmc WRITE_SIZE / FETCH_SIZE passes                -> HBM bytes per launch (per step = / K): shows that an
#   HBM-sized ring writes its algorithmic bytes and that a ring within reach of the 256 MiB Infinity Cache writes FEWER.
# usage: tools/collect_rollout_profiles.sh <outdir under gpurun_out> "cfg2 2" "cfg3 2" "cfg3 8" "cfg1 2" ...
set -u
R=$PWD
OUT=$R/${1:-gpurun_out/rollprof}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPECS=("$@")
for spec in "${SPECS[@]}"; do
  wl=${spec% *}; slots=${spec#* }; tag=${wl}_s${slots}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -- python3 $R/tools/rollout_target.py $wl $slots 64 10 > $OUT/${tag}_line.json 2> $OUT/${tag}_stats.err
  for ctr in WRITE_SIZE FETCH_SIZE; do
    PGX_PLACEMENT=0 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${tag}_$ctr -- python3 $R/tools/rollout_target.py $wl $slots 64 3 > $OUT/${tag}_${ctr}_line.json 2> $OUT/${tag}_$ctr.err
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/calib_$ctr -- $R/tools/hbm_peak > $OUT/calib_$ctr.txt 2> $OUT/calib_$ctr.err
done
cd $R
python3 tools/summarize_rollout_profiles.py $OUT "${SPECS[@]}"
find $OUT -name "*.csv" -size +1M -delete
