"""Per-kernel means of rocprofv3 --pmc counters: python tools/pmc_by_kernel.py <rocprof output dir> [kernel substring ...]
Counters with several instances (per TCC channel / XCC) are also shown as min / max over instances of the per-launch sum."""
import csv, glob, os, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
root, subs = sys.argv[1], sys.argv[2:]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    tot = defaultdict(lambda: defaultdict(float))   # (kernel, counter) -> dispatch -> sum over instances
    n_inst = defaultdict(int)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        tot[(k[:60], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, c), d in sorted(tot.items()):
        v = list(d.values())
        print(f"{k:60s} {c:44s} launches {len(v):4d} mean {sum(v) / len(v):16.1f} min {min(v):14.1f} max {max(v):14.1f}")
