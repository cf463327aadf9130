#!/bin/bash
# Diagnostic: what kind of box is this?  Clocks / power / temperature sampled WHILE the step kernel runs, next to the
# kernel's and the pure store stream's speed.
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power (W)\|junction\|(Sensor memory)" | sed 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.25; done ) > /tmp/probe_samples.txt &
SP=$!
python bench.py --steps 40000 --no-cpu-baseline > /tmp/probe_bench.json 2>/dev/null
kill $SP 2>/dev/null
sort -t'(' -k4 /tmp/probe_samples.txt | awk '/sclk/' | sort -u | tail -4
python - <<'PY'
import json
d=json.loads(open("/tmp/probe_bench.json").readline()); print("kernel_us %.1f" % (d["roofline"]["kernel_ms"]*1e3))
PY
tools/store_sweep | head -1
