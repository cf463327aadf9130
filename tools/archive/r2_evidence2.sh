#!/bin/bash
# round-2 evidence for DESIGN 4c: rollout vs step loop, split engines (free-running and joined), configs[1] timeline,
# rocprofv3 kernel stats of a rollout bench
mkdir -p gpurun_out/ev2
python tools/rollout_vs_step.py cfg1 cfg3 cfg2 cfg4 2>&1 | grep cfg > gpurun_out/ev2/rollout_vs_step.txt
python tools/split_streams.py cfg2 cfg3 cfg4 2>&1 | grep cfg > gpurun_out/ev2/split_streams.txt
echo "# JOIN=1 (device-side join of the parts after every step)" >> gpurun_out/ev2/split_streams.txt
JOIN=1 python tools/split_streams.py cfg2 cfg3 2>&1 | grep cfg >> gpurun_out/ev2/split_streams.txt
python tools/timeline_short.py cfg1 2>&1 | grep -v amdgpu > gpurun_out/ev2/timeline_cfg1.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev2/prof -- python3 $R/tools/rollout_vs_step.py cfg2 > /dev/null 2> $R/gpurun_out/ev2/prof.err
cd $R
f=$(find gpurun_out/ev2/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { head -1 "$f"; grep 'pgx::' "$f"; } > gpurun_out/ev2/rollout_cfg2_kernel_stats.csv
rm -rf gpurun_out/ev2/prof
cat gpurun_out/ev2/*.txt; cat gpurun_out/ev2/rollout_cfg2_kernel_stats.csv | cut -c1-220
