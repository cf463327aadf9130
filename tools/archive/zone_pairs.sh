#!/bin/bash
# Diagnostic: when the zone walk finds nothing, do the candidates pair with EACH OTHER?  (PGX_ZONE_PAIRS=1)
for i in 1 2 3; do
PGX_ZONE_PAIRS=1 PGX_DEBUG=1 python - <<'PY' 2>&1 | grep "READ\|pair scan\|pgx_buffers\]   \|spread" | tail -16
import torch
from pogema_amd.buffers import ZoneBuffers
p = ZoneBuffers((8192, 64, 3, 11, 11), torch.float32, "cuda:0", count=2)
print("spread", p.info["spread"], p.info["spacer_gib"], p.info["same_zone_us"], p.info["final_us"])
PY
done
