"""Diagnostic: does the map of fast / slow pairings change between two zone walks of ONE process?  Builds N pools one
after the other (each destroyed before the next), every walk run to the end of its budget (PGX_ZONE_SCAN=1), and prints
one line of probe times per pool.  If consecutive walks of a process see different maps, a failed walk is worth repeating.
    python tools/archive/zone_retry.py [pools=4] [budget_gib=136]"""
import os
import re
import subprocess
import sys

if os.environ.get("PGX_ZONE_SCAN_CHILD") != "1":
    env = dict(os.environ, PGX_ZONE_SCAN="1", PGX_DEBUG="1", PGX_ZONE_SCAN_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    pool = -1
    rows = {}
    for ln in p.stderr.splitlines():
        if ln.startswith("== pool"):
            pool = int(ln.split()[2])
            rows[pool] = []
        m = re.search(r"(\d+) GiB of spacers: candidate ([0-9.]+) us \(same-zone pair ([0-9.]+) us\)", ln)
        if m and pool >= 0:
            rows[pool].append((m.group(2), m.group(3)))
    for k, v in rows.items():
        print(f"pool {k}: same-zone pair {v[0][1] if v else '?'} us; candidates every 8 GiB [us]: " + " ".join(t for t, _ in v))
    if not rows:
        print(p.stderr[-1500:])
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc  # noqa: E402
import torch  # noqa: E402
from pogema_amd.buffers import ZoneBuffers  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 136.0
for k in range(n):
    print(f"== pool {k}", file=sys.stderr, flush=True)
    pool = ZoneBuffers((8192, 64, 3, 11, 11), torch.float32, "cuda:0", count=2, max_spacer_gib=budget)
    del pool
    gc.collect()
    torch.cuda.synchronize()
