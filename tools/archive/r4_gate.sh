#!/bin/bash
# round 4 experiment A: stream gate on the short launches (DESIGN.md 4 "Round 4")
O=gpurun_out/r4c; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python tools/ab_inproc.py cfg3 "" "PGX_GATE_NS=3000" "PGX_GATE_NS=4000" "PGX_GATE_NS=5000" "PGX_GATE_NS=6000" "PGX_GATE_NS=7000" "PGX_GATE_NS=8000" "PGX_GATE_NS=10000" > $O/gate_ab_cfg3.txt 2>&1
python tools/ab_inproc.py cfg1 "" "PGX_GATE_NS=3000" "PGX_GATE_NS=4000" "PGX_GATE_NS=5000" > $O/gate_ab_cfg1.txt 2>&1
python tools/ab_inproc.py cfg2 "" "PGX_GATE_NS=4000" "PGX_GATE_NS=6000" "PGX_GATE_NS=8000" > $O/gate_ab_cfg2.txt 2>&1
TL_STEP=2 python tools/wave_timeline.py cfg3 > $O/timeline_cfg3_nogate.txt 2>&1
PGX_GATE_NS=6000 TL_STEP=2 python tools/wave_timeline.py cfg3 > $O/timeline_cfg3_gate6.txt 2>&1
PGX_GATE_NS=8000 TL_STEP=2 python tools/wave_timeline.py cfg3 > $O/timeline_cfg3_gate8.txt 2>&1
tail -9 $O/gate_ab_cfg3.txt; tail -5 $O/gate_ab_cfg1.txt; tail -5 $O/gate_ab_cfg2.txt
