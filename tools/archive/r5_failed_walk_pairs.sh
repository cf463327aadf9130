#!/bin/bash
# one-off (round 5): in a process whose default walk fails, do the candidates pair with EACH OTHER?  Loops fresh processes
# until a walk fails (at most $1), then that process prints pgx_buffers' pair scan (PGX_ZONE_PAIRS).
N=${1:-14}
for i in $(seq 1 $N); do
  PGX_ZONE_PAIRS=1 python - <<'PY' 2> /tmp/pairs_err.txt
import torch, sys
from pogema_amd.buffers import ZoneBuffers
free, total = torch.cuda.mem_get_info(0)
zb = ZoneBuffers((8192, 64, 3, 11, 11), torch.float32, "cuda:0", count=3, max_spacer_gib=min(272.0, 0.5 * free / 2**30))
print("spread", zb.info["spread"], "candidates", zb.info["candidates"], "same_zone_us", zb.info["same_zone_us"], "final_us", zb.info["final_us"], flush=True)
sys.exit(0 if zb.info["spread"] else 3)
PY
  rc=$?
  echo "process $i rc=$rc"
  if [ $rc -eq 3 ]; then echo "---- failed walk in process $i: pair scan"; grep "pgx_buffers" /tmp/pairs_err.txt | tail -30; break; fi
done
