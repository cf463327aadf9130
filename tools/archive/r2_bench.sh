O=gpurun_out/r2_bench; mkdir -p $O
for wl in cfg2 cfg3 cfg4 cfg1; do
python bench.py --no-cpu-baseline --workload $wl --steps 500 > $O/$wl.json 2> $O/$wl.err
done
for f in cfg2 cfg3 cfg4 cfg1; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.3e'%d['value'], 'kernel_us %.2f'%(r['kernel_ms']*1e3), 'frac %.3f'%r['frac'], 'default_us', r['default_placement_kernel_ms'] and round(r['default_placement_kernel_ms']*1e3,2), d['config']['obs_buffers'])
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
