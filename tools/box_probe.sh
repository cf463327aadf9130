#!/bin/bash
# What kind of box is this lease?  (VERDICT r3 #3a)  Run BEFORE a benchmark on the same gpurun call and keep the output
# next to the bench line: partition modes (compute SPX/CPX..., memory NPS1/NPS4...), VRAM vendor / size, the KFD memory
# banks, firmware -- so that "zone" and "no-zone" boxes can be told apart by something other than timing.
#   tools/box_probe.sh [out.json]        default: gpurun_out/box_fingerprint.json
# bench.py embeds the sysfs part of this in every JSON line (`box`); this script adds what needs the smi tools.
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-gpurun_out/box_fingerprint.json}
mkdir -p "$(dirname "$OUT")"
TXT="${OUT%.json}.txt"
{
  echo "== date: $(date -u +%FT%TZ)  host: $(hostname)  kernel: $(uname -r)"
  echo "== rocm-smi partitions"
  timeout 60 rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^=\|^$" | head -20
  echo "== rocm-smi ids / vram / firmware"
  timeout 60 rocm-smi --showuniqueid --showmeminfo vram --showvbios --showbus 2>&1 | grep -v "^=\|^$" | head -30
  echo "== rocm-smi temperatures / power (idle: the probe runs before the benchmark)"
  timeout 60 rocm-smi --showtemp --showpower 2>&1 | grep -i "temperature\|power" | head -12
  echo "== rocm-smi clocks"
  timeout 60 rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk" | head -12
  if command -v amd-smi >/dev/null 2>&1; then
    echo "== amd-smi static (partition, vram, board)"
    timeout 90 amd-smi static --partition --vram --board --asic 2>&1 | head -80
  fi
  echo "== lscpu"
  lscpu 2>/dev/null | grep -i "model name\|^CPU(s)\|socket\|numa node(s)" | head -6
  echo "== /proc/meminfo"
  grep -i "memtotal\|hugepages_total" /proc/meminfo
} > "$TXT" 2>&1
python3 - "$OUT" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
fp = bench.box_fingerprint()
json.dump(fp, open(sys.argv[1], "w"), indent=1)
cards = fp["drm_cards"]
print("box:", fp["hostname"], "| GPUs visible:", sum(1 for c in cards if c.get("visible_to_this_process")), "|",
      "; ".join(f"{c['card']} {c['compute_partition']}/{c['memory_partition']} vram {c['vram_vendor']} {c['vram_total']} uid {c['unique_id']}" if "card" in c else str(c) for c in cards))
for n in fp["kfd_gpu_nodes"]:
    print(" kfd node", n["node"], {k: n[k] for k in ("num_xcc", "simd_count", "max_engine_clk_fcompute", "local_mem_size")},
          "banks:", [(b["heap_type"], b["size_in_bytes"], b["width"], b["mem_clk_max"]) for b in n["mem_banks"]])
PY
echo "-> $OUT, $TXT"
