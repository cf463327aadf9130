#!/bin/bash
O=gpurun_out/${1:-r4g}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
for i in 1 2 3; do python tools/archive/zone_scan.py all >> $O/zone_scan.txt 2>&1; done
python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --no-cpu-baseline --no-extras --placement-budget all > $O/bench_cfg2_budget_all.json 2> $O/bench_cfg2_budget_all.err
head -2 $O/box_probe.log; cat $O/zone_scan.txt
python tools/archive/fingerprint_table.py $O | tail -2 | cut -c1-230
