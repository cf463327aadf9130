#!/bin/bash
# Runs ON THE GPU BOX (under gpurun): rocprofv3 evidence for bench.py's workloads.
#   1. --kernel-trace --stats of the bench command (without the extra default-placement window, the secondary figures
#      and the CPU baseline, which launch the same kernel on other buffers / batches or burn host time) -> kernel_stats csv
#   2./3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md "HBM")           -> per-launch counters
#   4. counter calibration on tools/hbm_peak's known byte counts
#   5. (cfg2) one pass of the write-request stall counters -- TCC_EA0_WRREQ_DRAM_CREDIT_STALL, TCC_EA0_WRREQ_STALL, TCC_EA0_WRREQ --
#      WITH the product's placement (the walk's probe kernels are filtered out by kernel name): where the cycles of a
#      slow-tier lease go.  The summary is tagged with the tier the lease turned out to be
#      (`placement_tier`: "zone" = the walk spread the buffers over two zones, "nozone" = it found nothing): copy
#      cfg2_* to profiles/rN/cfg2_<tier>_* -- whichever tier the lease happens to be; nobody goes looking for one.
# usage: tools/collect_profiles.sh <outdir under gpurun_out> [workload ...]
set -u
R=$PWD
OUT=$R/${1:-gpurun_out/prof}; shift
WLS=${@:-cfg2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  extra="--workload $wl --no-default-placement --no-cpu-baseline --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${wl}_stats -- python3 $R/bench.py $extra > $OUT/${wl}_bench.json 2> $OUT/${wl}_stats.err
  for ctr in FETCH_SIZE WRITE_SIZE; do
    # PGX_PLACEMENT=0: no zone walk (its probe kernels and spacers are irrelevant for per-launch traffic)
    PGX_PLACEMENT=0 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${wl}_$ctr -- python3 $R/bench.py $extra --steps 20 --warmup 2 --windows 1 > /dev/null 2> $OUT/${wl}_$ctr.err
  done
  if [ "$wl" = cfg2 ]; then  # (the program itself directly after `--`; counters only with --kernel-trace)
    for ctr in TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ; do  # one counter per pass: an unknown name costs one pass only
      rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${wl}_STALL -- python3 $R/bench.py $extra --steps 20 --warmup 2 --windows 1 > $OUT/${wl}_stall_bench_$ctr.json 2> $OUT/${wl}_STALL_$ctr.err
    done
    cp $OUT/${wl}_stall_bench_TCC_EA0_WRREQ_DRAM_CREDIT_STALL.json $OUT/${wl}_stall_bench.json 2>/dev/null
  fi
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/calib_$ctr -- $R/tools/hbm_peak > $OUT/calib_$ctr.txt 2> $OUT/calib_$ctr.err
done
cd $R
python3 tools/summarize_profiles.py $OUT $WLS
find $OUT -name "*.csv" -size +1M -delete
ls $OUT
