O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab5.txt
python tools/ab_inproc.py cfg2 "PGX_FLAGS=0" "PGX_FLAGS=256" "PGX_FLAGS=0" "PGX_FLAGS=256" 2>&1 | grep -v amdgpu.ids >> $O/ab5.txt
python tools/ab_inproc.py cfg4 "PGX_FLAGS=0" "PGX_FLAGS=256" "PGX_FLAGS=0" "PGX_FLAGS=256" 2>&1 | grep -v amdgpu.ids >> $O/ab5.txt
cat $O/ab5.txt
