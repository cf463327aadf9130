#!/bin/bash
# Runs ON THE GPU BOX: SQ-level counters of the rollout loop (LDS bank conflicts, instruction mix, wave / busy cycles), one
# rocprofv3 --pmc pass per counter group (counters only with --kernel-trace; the program itself behind `--`).
# usage: tools/collect_sq_counters.sh <outdir under gpurun_out> "cfg1 2" "cfg3 2" ...
set -u
R=$PWD
OUT=$R/${1:-gpurun_out/sq}; shift
mkdir -p $OUT
SPECS=("$@")
cd /tmp && export TMPDIR=/tmp
for spec in "${SPECS[@]}"; do
  wl=${spec% *}; slots=${spec#* }; tag=${wl}_s${slots}
  g=0
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
    g=$((g+1))
    PGX_PLACEMENT=0 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${tag}_g$g -- python3 $R/tools/rollout_target.py $wl $slots 64 3 > /dev/null 2> $OUT/${tag}_g$g.err
  done
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
csv.field_size_limit(1 << 30)
res = {}
for path in sorted(glob.glob(os.path.join(out, "*_g*/**/*counter_collection.csv"), recursive=True)):
    tag = os.path.relpath(path, out).split(os.sep)[0].rsplit("_g", 1)[0]
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "rollout_kernel" not in k:
                continue
            d = res.setdefault(tag, {}).setdefault(row["Counter_Name"], [0.0, set()])
            d[0] += float(row["Counter_Value"])
            d[1].add(row.get("Dispatch_Id"))
summary = {tag: {c: v[0] / max(1, len(v[1])) for c, v in cs.items()} for tag, cs in res.items()}
for tag, cs in summary.items():
    if cs.get("SQ_LDS_IDX_ACTIVE"):
        cs["lds_bank_conflict_cycles_per_active_cycle"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
json.dump(summary, open(os.path.join(out, "sq_counters_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
