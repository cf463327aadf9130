"""Diagnostic: the zone walk of pgx_buffers_create run to the END of its budget instead of stopping at the first faster
candidate (PGX_ZONE_SCAN=1), one probe time per 8 GiB of spacers: the map of "which stretches of the allocation order pair
fast with where this process's first allocation landed".  One line per process; run it several times per lease -- the
starting point differs from process to process on the same GPU (profiles/r4/box_fingerprints.md).
    python tools/archive/zone_scan.py [budget_gib=all]"""
import os
import re
import subprocess
import sys

if os.environ.get("PGX_ZONE_SCAN_CHILD") != "1":
    env = dict(os.environ, PGX_ZONE_SCAN="1", PGX_DEBUG="1", PGX_ZONE_SCAN_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    times = re.findall(r"(\d+) GiB of spacers: candidate ([0-9.]+) us \(same-zone pair ([0-9.]+) us\)", p.stderr)
    if not times:
        print("no scan output:", p.stderr[-800:])
        sys.exit(1)
    print(f"same-zone pair {times[0][2]} us; candidates every 8 GiB from 8 to {times[-1][0]} GiB [us]: " + " ".join(t for _, t, _ in times))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pogema_amd.buffers import ZoneBuffers  # noqa: E402

budget = sys.argv[1] if len(sys.argv) > 1 else "all"
pool = ZoneBuffers((8192, 64, 3, 11, 11), torch.float32, "cuda:0", count=2, max_spacer_gib="all" if budget == "all" else float(budget))
