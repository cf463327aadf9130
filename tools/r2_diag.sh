O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab.txt
for wl in cfg3 cfg1 cfg2 cfg4; do
python tools/ab_inproc.py $wl "PGX_FLAGS=0" "PGX_FLAGS=1" "PGX_FLAGS=16" "PGX_FLAGS=32" "PGX_FLAGS=0" 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
