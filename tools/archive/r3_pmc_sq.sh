#!/bin/bash
# round 3: instruction-mix / LDS / issue counters of the step kernel (separate --pmc passes; --kernel-trace only, as gpurun requires)
R=$PWD; O=$R/gpurun_out/r3_pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3; do
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PGX_PLACEMENT=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${wl}_set$i -- python3 $R/bench.py --workload $wl --no-default-placement --no-cpu-baseline --no-extras --steps 20 --warmup 2 --windows 1 > /dev/null 2> $O/${wl}_set$i.err
  python3 $R/tools/pmc_by_kernel.py $O/${wl}_set$i step_kernel | tee -a $O/${wl}_sq_counters.txt
done
done
find $O -name "*.csv" -size +1M -delete
