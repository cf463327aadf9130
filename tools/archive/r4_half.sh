O=gpurun_out/r4q; mkdir -p $O
bash tools/box_probe.sh $O/box.json > /dev/null 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_rollout_gpu.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/parity.log
for dt in bfloat16 float16 uint8 float32; do python bench.py --obs-dtype $dt --no-cpu-baseline --no-default-placement > $O/bench_cfg2_$dt.json 2> $O/bench_$dt.err; done
python bench.py --workload cfg4 --obs-dtype bfloat16 --no-cpu-baseline --no-default-placement --no-extras > $O/bench_cfg4_bfloat16.json 2>> $O/bench_bfloat16.err
python bench.py --workload cfg3 --obs-dtype bfloat16 --no-cpu-baseline --no-default-placement --no-extras > $O/bench_cfg3_bfloat16.json 2>> $O/bench_bfloat16.err
tail -2 $O/parity.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try: d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e: print(sys.argv[1], "ERR", e); sys.exit(0)
r = d["roofline"]
print(f"{sys.argv[1].split('/')[-1]:30s} {d['ms_per_step']*1e3:7.1f} us/step  {d['value']:.3e} agent-steps/s  achieved {r['achieved']:.0f} GB/s frac {r['frac']:.3f}  bytes/agent-step {r['algorithmic_bytes_per_agent_step']:.0f}")
PY
done
