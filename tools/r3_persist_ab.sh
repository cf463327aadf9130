#!/bin/bash
# round 3: persistent producer/consumer workgroups (step_persist_kernel) against the one-workgroup-per-environment launch
out=gpurun_out/r3j; mkdir -p $out
timeout 300 python -m pytest tests/test_persist_gpu.py -x -q 2>&1 | tail -2
timeout 400 python tools/ab_inproc.py cfg2 "PGX_PERSIST=0" "PGX_PERSIST=1" "PGX_PERSIST=1,PGX_PERSIST_PER_CU=6" "PGX_PERSIST=1,PGX_PERSIST_PER_CU=4" "PGX_PERSIST=1,PGX_PSHARE=0" "PGX_PERSIST=1,PGX_PSHARE=33" "PGX_PERSIST=0" > $out/persist_ab_cfg2.txt 2>&1
tail -9 $out/persist_ab_cfg2.txt
PGX_PERSIST=1 timeout 120 python tools/wave_timeline.py cfg2 > $out/timeline_cfg2_persist.txt 2>&1
tail -40 $out/timeline_cfg2_persist.txt
