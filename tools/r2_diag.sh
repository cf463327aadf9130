O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab2.txt
python tools/ab_inproc.py cfg3 "PGX_FLAGS=16" "PGX_FLAGS=16,PGX_LDS_MIN=10000" "PGX_FLAGS=16,PGX_LDS_MIN=20000" "PGX_FLAGS=16,PGX_LDS_MIN=40000" "PGX_FLAGS=16,PGX_EPW=2" "PGX_FLAGS=16,PGX_EPW=2,PGX_LDS_MIN=20000" "PGX_FLAGS=16,PGX_EPW=4" "PGX_FLAGS=16,PGX_EPW=4,PGX_LDS_MIN=40000" "PGX_FLAGS=16" 2>&1 | grep -v amdgpu.ids >> $O/ab2.txt
python tools/ab_inproc.py cfg1 "PGX_FLAGS=16" "PGX_FLAGS=16,PGX_LDS_MIN=20000" "PGX_FLAGS=16,PGX_LDS_MIN=40000"  "PGX_FLAGS=16,PGX_EPW=2" "PGX_FLAGS=16" 2>&1 | grep -v amdgpu.ids >> $O/ab2.txt
python tools/ab_inproc.py cfg2 "PGX_FLAGS=16" "PGX_FLAGS=16,PGX_LDS_MIN=10000" "PGX_FLAGS=16,PGX_LDS_MIN=20000" "PGX_FLAGS=16,PGX_STAGGER=0" "PGX_FLAGS=16,PGX_STAGGER=0,PGX_LDS_MIN=10000" "PGX_FLAGS=16" 2>&1 | grep -v amdgpu.ids >> $O/ab2.txt
cat $O/ab2.txt
