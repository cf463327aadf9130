"""Condenses the rocprofv3 output of tools/collect_profiles.sh into small files fit for profiles/:
<wl>_kernel_stats.csv (our kernels only), <wl>_pmc_summary.json (per-launch FETCH_SIZE / WRITE_SIZE of
pgx::step_kernel, corrected as MI355X_MICROARCH.md prescribes, with the calibration that justifies it)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
wls = sys.argv[2:] or ["cfg2"]
csv.field_size_limit(1 << 30)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return hits[0] if hits else None


def counter_per_launch(dirname, counter, kernel_substr):
    paths = sorted(glob.glob(os.path.join(out, f"{dirname}/**/*counter_collection.csv"), recursive=True))
    if not paths:
        return None
    per_dispatch = {}
    for path in paths:  # (several passes may share a directory: one file per process)
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter or kernel_substr not in row.get("Kernel_Name", ""):
                    continue
                key = (path, row.get("Dispatch_Id"))
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
    if not per_dispatch:
        return None
    vals = list(per_dispatch.values())
    return {"mean": sum(vals) / len(vals), "launches": len(vals), "min": min(vals), "max": max(vals)}


calib = {}
for ctr, kern, known in (("WRITE_SIZE", "fill_chunk_kernel", 761266176), ("FETCH_SIZE", "copy_kernel", 761266176)):
    c = counter_per_launch(f"calib_{ctr}", ctr, kern)
    if c:
        calib[ctr] = {"kernel": kern, "known_bytes": known, "counter_kb_mean": c["mean"],
                      "ratio_counter_bytes_to_known": c["mean"] * 1024.0 / known}

for wl in wls:
    stats = find(f"{wl}_stats/**/*kernel_stats.csv")
    if stats:
        with open(stats) as f, open(os.path.join(out, f"{wl}_kernel_stats.csv"), "w") as g:
            for i, line in enumerate(f):
                if i == 0 or line.startswith('"pgx::') or line.startswith('"void pgx::'):
                    g.write(line)
    summary = {"workload": wl, "kernel": "pgx::step_kernel", "calibration": calib}
    fetch = counter_per_launch(f"{wl}_FETCH_SIZE", "FETCH_SIZE", "step_kernel")
    write = counter_per_launch(f"{wl}_WRITE_SIZE", "WRITE_SIZE", "step_kernel")
    if fetch and write:
        fr = calib.get("FETCH_SIZE", {}).get("ratio_counter_bytes_to_known", 0.5)
        wr = calib.get("WRITE_SIZE", {}).get("ratio_counter_bytes_to_known", 1.0)
        summary.update({
            "FETCH_SIZE_kb_mean": fetch["mean"], "WRITE_SIZE_kb_mean": write["mean"], "launches": fetch["launches"],
            # guide: FETCH_SIZE reports 1/2 of a 16-B/lane stream on gfx950 -> x2; WRITE_SIZE as calibrated
            "hbm_bytes_per_launch": fetch["mean"] * 1024.0 / (fr if 0.4 < fr < 0.6 else 0.5)
                                    + write["mean"] * 1024.0 / (wr if 0.9 < wr < 1.1 else 1.0),
        })
    bench = os.path.join(out, f"{wl}_bench.json")
    if os.path.exists(bench):
        try:
            with open(bench) as f:
                summary["bench_line"] = json.loads(f.readline())
            pl = (summary["bench_line"].get("roofline") or {}).get("placement") or {}
            # "nozone" only after a walk that really went far and found nothing; short launches (< 128 MiB: no walk) and
            # walks that stopped early are "unplaced"
            summary["placement_tier"] = ("zone" if pl.get("spread") else
                                         "nozone" if (pl.get("walk_candidates") or 0) >= 8 else "unplaced")
        except Exception:
            pass
    # write-request stall counters of the step kernel (one pass, product placement): sums over the TCC channel instances
    # per launch, and the stalled share of the write requests' cycles
    stall = {}
    for ctr in ("TCC_EA0_WRREQ_DRAM_CREDIT_STALL", "TCC_EA0_WRREQ_STALL", "TCC_EA0_WRREQ"):
        c = counter_per_launch(f"{wl}_STALL", ctr, "step_kernel")
        if c:
            stall[ctr] = c
    if stall:
        sb = os.path.join(out, f"{wl}_stall_bench.json")
        line = None
        if os.path.exists(sb):
            try:
                with open(sb) as f:
                    line = json.loads(f.readline())
            except Exception:
                line = None
        rf = (line or {}).get("roofline") or {}
        summary["write_stalls"] = {
            "per_launch": stall,
            "dram_credit_stall_cycles_per_wrreq": (stall["TCC_EA0_WRREQ_DRAM_CREDIT_STALL"]["mean"] / stall["TCC_EA0_WRREQ"]["mean"])
            if "TCC_EA0_WRREQ" in stall and "TCC_EA0_WRREQ_DRAM_CREDIT_STALL" in stall and stall["TCC_EA0_WRREQ"]["mean"] else None,
            "wrreq_stall_cycles_per_wrreq": (stall["TCC_EA0_WRREQ_STALL"]["mean"] / stall["TCC_EA0_WRREQ"]["mean"])
            if "TCC_EA0_WRREQ" in stall and "TCC_EA0_WRREQ_STALL" in stall and stall["TCC_EA0_WRREQ"]["mean"] else None,
            "pass_kernel_ms": rf.get("kernel_ms"), "pass_spread": (rf.get("placement") or {}).get("spread"),
            "pass_box_store_stream_gbs": rf.get("box_store_stream_gbs"),
            "note": "counters are summed over all TCC channel instances per launch of pgx::step_kernel; collected in their own "
                    "rocprofv3 pass (--kernel-trace --pmc only) with the product's placement, so pass_spread says which tier "
                    "THIS pass ran on",
        }
    with open(os.path.join(out, f"{wl}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(wl, {k: summary.get(k) for k in ("FETCH_SIZE_kb_mean", "WRITE_SIZE_kb_mean", "hbm_bytes_per_launch")})
print("calibration", calib)
