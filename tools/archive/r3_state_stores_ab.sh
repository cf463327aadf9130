#!/bin/bash
# round 3: when the small per-step result stores are issued (PGX_STATE_STORES = 0 at once | 1 after the barrier | 2 after the stream)
out=gpurun_out/r3d; mkdir -p $out
PGX_STATE_STORES=2 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_rollout_gpu.py tests/test_fullsize_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $out/pytest_ss2.log 2>&1; echo "ss2 rc=$?"; tail -2 $out/pytest_ss2.log
for wl in cfg3 cfg1 cfg2 cfg4; do
  timeout 400 python tools/ab_inproc.py $wl "PGX_STATE_STORES=0" "PGX_STATE_STORES=1" "PGX_STATE_STORES=2" "PGX_STATE_STORES=0" "PGX_STATE_STORES=2" > $out/ss_ab_$wl.txt 2>&1
  tail -6 $out/ss_ab_$wl.txt
done
PGX_STATE_STORES=2 timeout 200 python tools/timeline_short.py cfg3 > $out/timeline_short_cfg3_ss2.txt 2>&1
