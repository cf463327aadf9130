#!/bin/bash
# round 4 experiment B: overlapping-word LDS bitmaps + packed row stores in the single-wave kernels (parity, then A/B)
O=gpurun_out/${1:-r4e}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_rollout_gpu.py tests/test_fullsize_gpu.py tests/test_spec_vectors_gpu.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
P=pogema_amd/libpogema_amd_plain.so; K=pogema_amd/libpogema_amd_packed.so
python tools/ab_inproc.py cfg3 "" "PGX_LIB=$P" "PGX_LIB=$K" "" "PGX_LIB=$P" "PGX_LIB=$K" > $O/ovl_ab_cfg3.txt 2>&1
python tools/ab_inproc.py cfg1 "" "PGX_LIB=$P" "PGX_LIB=$K" "" "PGX_LIB=$P" "PGX_LIB=$K" > $O/ovl_ab_cfg1.txt 2>&1
python tools/ab_inproc.py cfg2 "PGX_LIB=$P" "PGX_LIB=$K" "PGX_LIB=$P" "PGX_LIB=$K" > $O/ovl_ab_cfg2.txt 2>&1
python tools/ab_inproc.py a32 "" "PGX_LIB=$P" "PGX_LIB=$K" > $O/ovl_ab_a32.txt 2>&1
tail -2 $O/parity.log; for f in cfg3 cfg1 cfg2 a32; do grep -h "spread" $O/ovl_ab_$f.txt | cut -c1-60; tail -7 $O/ovl_ab_$f.txt; done
