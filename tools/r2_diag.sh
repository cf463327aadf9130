O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab4.txt
python tools/ab_inproc.py cfg3 "PGX_TEAM=1" "PGX_TEAM=2" "PGX_TEAM=4" "PGX_TEAM=8" "PGX_TEAM=16" "PGX_TEAM=4,PGX_EPW=2" "PGX_TEAM=8,PGX_EPW=4" "PGX_TEAM=1" 2>&1 | grep -v amdgpu.ids >> $O/ab4.txt
python tools/ab_inproc.py cfg1 "PGX_TEAM=1" "PGX_TEAM=2" "PGX_TEAM=4" "PGX_TEAM=8" "PGX_TEAM=1" 2>&1 | grep -v amdgpu.ids >> $O/ab4.txt
python tools/ab_inproc.py cfg2 "PGX_TEAM=1" "PGX_TEAM=2" "PGX_TEAM=4" "PGX_TEAM=8" "PGX_TEAM=1" 2>&1 | grep -v amdgpu.ids >> $O/ab4.txt
cat $O/ab4.txt
