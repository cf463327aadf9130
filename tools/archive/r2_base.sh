set -u
O=gpurun_out/r2_base; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
for wl in cfg1 cfg3 cfg2; do for ad in int64 int8; do
python bench.py --workload $wl --action-dtype $ad --no-cpu-baseline --steps 2000 > $O/bench_${wl}_${ad}.json 2>&1
done; done
python tools/state_phase_timeline.py cfg3 > $O/timeline_cfg3.txt 2>&1
python tools/state_phase_timeline.py cfg1 > $O/timeline_cfg1.txt 2>&1
tools/hbm_peak > $O/hbm_peak.txt 2>&1
tail -3 $O/pytest.txt; grep -h -o '"kernel_ms": [0-9.]*' $O/bench_*.json
