O=gpurun_out/r2_diag; mkdir -p $O; rm -f $O/ab7.txt
for wl in cfg2 cfg3 cfg1; do
python tools/ab_inproc.py $wl "PGX_STORE=sc1" "PGX_STORE=nt" "PGX_STORE=plain" "PGX_STORE=sc1" 2>&1 | grep -v amdgpu.ids >> $O/ab7.txt
done
cat $O/ab7.txt
