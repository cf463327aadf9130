#!/bin/bash
# One lease's contribution to profiles/rN/box_fingerprints.md: which GPU, two whole-device zone scans (two processes), and
# the headline bench under the product-default walk budget.   usage: tools/lease_survey.sh <name under gpurun_out>
O=gpurun_out/${1:-survey}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
for i in 1 2; do python tools/zone_scan.py all >> $O/zone_scan.txt 2>&1; done
python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
head -1 $O/box_probe.log | cut -c1-120; cut -c1-200 $O/zone_scan.txt; python tools/fingerprint_table.py $O | tail -1 | cut -c1-230
