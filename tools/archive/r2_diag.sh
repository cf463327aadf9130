O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab8.txt
PGX_SEQ=3 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rollout_parity" 2>&1 | tail -1
python tools/ab_inproc.py cfg3 "PGX_SEQ=1" "PGX_SEQ=2" "PGX_SEQ=4" "PGX_SEQ=2,PGX_EPW=2" "PGX_SEQ=4,PGX_EPW=2" "PGX_SEQ=8" "PGX_SEQ=1" 2>&1 | grep -v amdgpu.ids >> $O/ab8.txt
python tools/ab_inproc.py cfg2 "PGX_SEQ=1" "PGX_SEQ=2" "PGX_SEQ=4" "PGX_SEQ=1" 2>&1 | grep -v amdgpu.ids >> $O/ab8.txt
python tools/ab_inproc.py cfg1 "PGX_SEQ=1" "PGX_SEQ=2" "PGX_SEQ=1" 2>&1 | grep -v amdgpu.ids >> $O/ab8.txt
cat $O/ab8.txt
