#!/usr/bin/env python3
"""Registers / scratch / occupancy of every step and rollout kernel instance (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel.  `python tools/resource_usage.py [substring]`"""
import re
import subprocess
import sys

out = subprocess.run(["make", "-C", "pogema_amd/csrc", "resource-usage"], capture_output=True, text=True)
rows, cur = [], None
for ln in (out.stdout + out.stderr).splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).split()[0].replace("Total", "")] = int(m.group(2))
want = sys.argv[1] if len(sys.argv) > 1 else "kernel"
for r in rows:
    n = r["name"]
    m = re.match(r"_ZN3pgx\d+(\w+?)ILi(\d+)ELb(\d)ELb(\d)ELb(\d)(?:ELb(\d))?E", n)
    label = (f"{m.group(1)}{'_big' if m.group(5) == '1' else ''}{'_pair' if m.group(6) == '1' else ''}"
             f"<G={m.group(2)},MW={m.group(3)},P16={m.group(4)}>") if m else n[:50]
    if want in label:
        print(f"{label:40s} sgpr {r.get('SGPRs', '?'):>3} vgpr {r.get('VGPRs', '?'):>3} scratch {r.get('ScratchSize', '?'):>3} occ {r.get('Occupancy', '?')}")
