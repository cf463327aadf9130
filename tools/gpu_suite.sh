#!/bin/bash
# the whole GPU suite without -x (all failures at once), output under gpurun_out/$1
O=gpurun_out/${1:-r4b}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python -m pytest tests -q -m gpu -s > $O/gpu_suite.log 2>&1; echo "suite rc=$?" | tee -a $O/gpu_suite.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.log | tail -30
