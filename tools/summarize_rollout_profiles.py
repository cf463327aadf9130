"""Condenses tools/collect_rollout_profiles.sh's rocprofv3 output: rollout_<tag>_kernel_stats.csv (our kernels) and
rollout_pmc_summary.json (per launch and per step: WRITE_SIZE / FETCH_SIZE of pgx::rollout_kernel, corrected as in
tools/summarize_profiles.py, next to the algorithmic bytes and the HIP-event time of the un-profiled stats run)."""
import csv, glob, json, os, sys
out, specs = sys.argv[1], sys.argv[2:]
csv.field_size_limit(1 << 30)


def counter_per_launch(dirname, counter, kernel_substr):
    per = {}
    for path in sorted(glob.glob(os.path.join(out, f"{dirname}/**/*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                    key = (path, row.get("Dispatch_Id"))
                    per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    vals = list(per.values())
    return {"mean": sum(vals) / len(vals), "launches": len(vals)} if vals else None


calib = {}
for ctr, kern, known in (("WRITE_SIZE", "fill_chunk_kernel", 761266176), ("FETCH_SIZE", "copy_kernel", 761266176)):
    c = counter_per_launch(f"calib_{ctr}", ctr, kern)
    if c:
        calib[ctr] = c["mean"] * 1024.0 / known
summary = {"calibration_counter_bytes_to_known": calib, "runs": []}
for spec in specs:
    wl, slots = spec.split()
    tag = f"{wl}_s{slots}"
    hits = sorted(glob.glob(os.path.join(out, f"{tag}_stats/**/*kernel_stats.csv"), recursive=True))
    if hits:
        with open(hits[0]) as f, open(os.path.join(out, f"rollout_{tag}_kernel_stats.csv"), "w") as g:
            for i, line in enumerate(f):
                if i == 0 or "pgx::" in line:
                    g.write(line)
    rec = {"tag": tag}
    try:
        rec["stats_run"] = json.loads(open(os.path.join(out, f"{tag}_line.json")).read().strip().splitlines()[-1])
    except Exception as exc:  # noqa: BLE001
        rec["stats_run"] = {"error": repr(exc)}
    w = counter_per_launch(f"{tag}_WRITE_SIZE", "WRITE_SIZE", "rollout_kernel")
    f_ = counter_per_launch(f"{tag}_FETCH_SIZE", "FETCH_SIZE", "rollout_kernel")
    K = rec["stats_run"].get("steps_per_launch", 64)
    if w and f_:
        wr, fr = calib.get("WRITE_SIZE", 1.0), calib.get("FETCH_SIZE", 0.5)
        wb = w["mean"] * 1024.0 / (wr if 0.9 < wr < 1.1 else 1.0)
        fb = f_["mean"] * 1024.0 / (fr if 0.4 < fr < 0.6 else 0.5)
        rec.update(hbm_write_bytes_per_step=wb / K, hbm_fetch_bytes_per_step=fb / K, hbm_bytes_per_step=(wb + fb) / K,
                   launches_counted=w["launches"],
                   written_over_algorithmic=(wb / K) / rec["stats_run"].get("algorithmic_bytes_per_step", float("nan")))
    summary["runs"].append(rec)
    print(tag, {k: rec.get(k) for k in ("hbm_write_bytes_per_step", "written_over_algorithmic")}, rec["stats_run"].get("us_per_step_hip_events"))
json.dump(summary, open(os.path.join(out, "rollout_pmc_summary.json"), "w"), indent=1)
