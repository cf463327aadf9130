#!/bin/bash
# Runs ON THE GPU BOX (under gpurun): rocprofv3 evidence for bench.py's workloads.
#   1. --kernel-trace --stats of the exact default bench command               -> kernel_stats csv
#   2./3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md "HBM") -> per-launch counters
#   4. counter calibration on tools/hbm_peak's known byte counts
# usage: tools/collect_profiles.sh <outdir under gpurun_out> [workload ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=${1:-gpurun_out/prof}; shift
WLS=${@:-cfg2}
mkdir -p $OUT
for wl in $WLS; do
  extra="--workload $wl"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_stats -- python3 bench.py $extra > $OUT/${wl}_bench.json 2> $OUT/${wl}_stats.err
  for ctr in FETCH_SIZE WRITE_SIZE; do
    # PGX_PLACEMENT=0: no placement-probe launches (they are MODE_OBSERVE launches of the same kernel and would dilute
    # the per-step counter means)
    PGX_PLACEMENT=0 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${wl}_$ctr -- python3 bench.py $extra --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/${wl}_$ctr.err
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/calib_$ctr -- tools/hbm_peak > $OUT/calib_$ctr.txt 2> $OUT/calib_$ctr.err
done
python3 tools/summarize_profiles.py $OUT $WLS
