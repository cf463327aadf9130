for wl in cfg3 cfg1; do for nb in 2 1; do
python bench.py --no-cpu-baseline --no-default-placement --workload $wl --buffers $nb --steps 1000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wl buffers $nb kernel_us %.2f frac %.3f'%(r['kernel_ms']*1e3, r['frac']), d['config']['obs_buffers'][:40])"
done; done
