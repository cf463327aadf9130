#!/usr/bin/env python3
"""One line per (gpurun lease, process): which GPU it was (tools/box_probe.sh) and how the engine's zone walk went on it
(bench.py's roofline.placement, the `buffers:` line of tools/ab_inproc.py).  python tools/fingerprint_table.py gpurun_out/r4*
-> markdown on stdout (kept as profiles/r4/box_fingerprints.md): lets "zone" and "no-zone" boxes be told apart by
something other than timing (VERDICT r3 #3a)."""
import ast
import glob
import json
import os
import re
import sys

rows = []
for d in sorted(sys.argv[1:]):
    txt = os.path.join(d, "box.txt")
    if not os.path.exists(txt):
        continue
    t = open(txt).read()

    def grab(pat):
        m = re.search(pat, t)
        return m.group(1).strip() if m else "?"

    box = dict(uid=grab(r"Unique ID: (0x[0-9a-fA-F]+)"), oam=grab(r"OAM_ID: (\d+)"), model=grab(r"MODEL_NUMBER: (\S+)"),
               serial=grab(r"PRODUCT_SERIAL: (\S+)"), vbios=grab(r"VBIOS version: (\S+)"), bus=grab(r"PCI Bus: (\S+)"),
               part=grab(r"Compute Partition: (\S+)") + "/" + grab(r"Memory Partition: (\S+)"), vram=grab(r"VENDOR: (\S+)"))
    for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
        try:
            line = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            continue
        r, pl = line["roofline"], line["roofline"].get("placement") or {}
        rows.append((d, box, os.path.basename(f)[6:-5], str(pl.get("policy", "?")).split(":")[0], pl.get("budget_gib"), pl.get("spread"),
                     pl.get("walk_candidates"), pl.get("spacer_gib_held"), pl.get("probe_same_zone_us"), pl.get("probe_as_placed_us"),
                     round(r["kernel_ms"] * 1e3, 1), round((r.get("default_placement_kernel_ms") or 0) * 1e3, 1), round(r["frac"], 3)))
    for f in sorted(glob.glob(os.path.join(d, "*ab*.txt"))):
        for ln in open(f):
            if ln.startswith("buffers: {"):
                try:
                    pl = ast.literal_eval(ln[len("buffers: "):].strip())
                except Exception:  # noqa: BLE001
                    continue
                if "same_zone_us" not in pl:
                    continue
                rows.append((d, box, os.path.basename(f)[:-4], str(pl.get("policy", "?")).split(":")[0], pl.get("budget_gib"), pl.get("spread"),
                             pl.get("candidates"), pl.get("spacer_gib"), pl.get("same_zone_us"), pl.get("final_us"), None, None, None))
print("| lease | GPU unique id | OAM | board model | partitions | VRAM | process | walk policy | budget GiB | spread | candidates | spacers held GiB | "
      "probe same-zone us | probe as placed us | step kernel us | same kernel, torch-placed buffers us | frac of 8 TB/s |")
print("|" + "---|" * 17)
for d, b, name, pol, bud, spread, cand, held, same, final, k, kd, frac in rows:
    print(f"| {os.path.basename(d)} | {b['uid']} | {b['oam']} | {b['model']} | {b['part']} | {b['vram']} | {name} | {pol} | {bud} | {spread} | {cand} | "
          f"{held} | {same} | {final} | {k if k is not None else ''} | {kd if kd else ''} | {frac if frac is not None else ''} |")
