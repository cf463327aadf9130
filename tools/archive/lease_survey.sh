#!/bin/bash
# One lease's contribution to profiles/rN/box_fingerprints.md: which GPU, two whole-device zone scans (two processes), and
# the headline bench under the product-default walk budget.   usage: tools/lease_survey.sh <name under gpurun_out>
O=gpurun_out/${1:-survey}; mkdir -p $O
bash tools/box_probe.sh $O/box.json > $O/box_probe.log 2>&1
for i in 1 2; do python tools/archive/zone_scan.py all >> $O/zone_scan.txt 2>&1; done
# temperatures / power / clocks sampled WHILE the benchmark runs (a child of this shell, not of a GPU process)
( while true; do rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -i "memory) (C)\|junction) (C)\|Socket Graphics Package Power\|mclk\|fclk\|sclk" | sed 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.4; done ) > $O/samples.txt 2>/dev/null &
SAMPLER=$!
python bench.py --no-cpu-baseline --no-extras --steps 20000 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
kill $SAMPLER 2>/dev/null
python3 - $O/samples.txt <<'PY' | tee $O/samples_summary.txt
import re, sys
mem, junc, pw, mclk = [], [], [], []
for ln in open(sys.argv[1]):
    for pat, dst in ((r"memory\) \(C\): ([0-9.]+)", mem), (r"junction\) \(C\): ([0-9.]+)", junc), (r"Power \(W\): ([0-9.]+)", pw), (r"mclk clock level: \d+: \((\d+)Mhz", mclk)):
        m = re.search(pat, ln)
        if m: dst.append(float(m.group(1)))
fmt = lambda v: f"max {max(v):.0f} median {sorted(v)[len(v)//2]:.0f}" if v else "n/a"
print(f"under load ({len(mem)} samples): HBM temperature {fmt(mem)} C; junction {fmt(junc)} C; package power {fmt(pw)} W; mclk {fmt(mclk)} MHz")
PY
head -1 $O/box_probe.log | cut -c1-120; cut -c1-200 $O/zone_scan.txt; python tools/archive/fingerprint_table.py $O | tail -1 | cut -c1-230
