#!/bin/bash
O=gpurun_out/${1:-r4j}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python tools/soak.py ${2:-6000} ${3:-96} > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/soak.txt
bash tools/fuzz_campaign.sh ${4:-6000} ${5:-7500} 100 $O/fuzz_campaign.txt > /dev/null 2>&1
tail -4 $O/soak.txt; tail -3 $O/fuzz_campaign.txt
