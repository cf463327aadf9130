#!/bin/bash
# round 3: prologue experiments -- PGX_FLAGS bit 13 (8192) raised priority during the prologue, bit 14 (16384) state stores after the barrier
out=gpurun_out/r3c; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_default.log 2>&1; echo "default rc=$?"; tail -2 $out/pytest_default.log
PGX_FLAGS=24576 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_rollout_gpu.py tests/test_fullsize_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $out/pytest_late.log 2>&1; echo "late rc=$?"; tail -2 $out/pytest_late.log
for wl in cfg2 cfg3 cfg4; do
  timeout 400 python tools/ab_inproc.py $wl "PGX_FLAGS=0" "PGX_FLAGS=8192" "PGX_FLAGS=16384" "PGX_FLAGS=24576" "PGX_FLAGS=0" > $out/prologue_ab_$wl.txt 2>&1
  tail -6 $out/prologue_ab_$wl.txt
done
