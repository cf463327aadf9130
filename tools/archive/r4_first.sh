#!/bin/bash
# round 4, first GPU call: fingerprint, the new tests, the whole GPU suite, default bench
O=gpurun_out/r4a; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
python -m pytest tests/test_sharded_gpu.py tests/test_bench_rehearsal_gpu.py tests/test_integration_doc_gpu.py tests/test_buffers_gpu.py -x -q -s -m gpu > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?" | tee -a $O/gpu_suite.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python bench.py --workload cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg1 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
tail -3 $O/new_tests.log; tail -3 $O/gpu_suite.log
