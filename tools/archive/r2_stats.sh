O=$PWD/gpurun_out/r2_stats; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for wl in cfg1 cfg3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${wl} -- python3 $R/bench.py --workload $wl --no-cpu-baseline --steps 1000 --windows 2 --no-default-placement > $O/${wl}_bench.json 2> $O/${wl}.err
grep -h "step_kernel" $O/${wl}/*/*kernel_stats.csv | cut -c1-160
python3 - $O/${wl}_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('bench kernel_us(launch-to-launch) %.2f'%(d['roofline']['kernel_ms']*1e3))
PY
done
find $O -name "*.csv" -size +1M -delete
