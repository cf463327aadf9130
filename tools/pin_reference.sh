#!/bin/bash
# One command from "parity unpinned" to a pinned contract -- run where the REAL `pogema` is importable:
#
#     tools/pin_reference.sh                 fixtures -> tests/golden/, then the checks below
#     tools/pin_reference.sh --out DIR       fixtures somewhere else (the stand-in rehearsal of tests/test_golden_pipeline.py)
#     tools/pin_reference.sh --limit N       first N cases only;   --geoms 0,3   only these geometries of gen_golden.py
#
#   1. tools/gen_golden.py            drives the importable `pogema` over the seeded cases: reference_*.npz (inputs, expected
#                                      outputs, the occupancy array, final metrics) + reference_probes.json
#   2. tests/pin_semantics.py         brute-forces the 2^4 positions of the semantics switches with the Python oracle over
#                                      those fixtures and prints which combination(s) make ALL of them pass
#   3. pytest tests/test_golden_reference.py -m "not gpu"     oracle + numpy-stream generator against every fixture under
#                                      the product's DEFAULT semantics (on a GPU box add:  -m gpu  for the engine)
# Exit code: 0 iff a passing combination exists AND the default-semantics tests pass.  When (2) passes but (3) fails, the
# report names the switch to flip (pogema_amd/semantics.py defaults, or PGX_SEMANTICS=... for one run).
set -u
cd "$(dirname "$0")/.." || exit 1
OUT=tests/golden; LIMIT=0; GEOMS=""
while [ $# -gt 0 ]; do
  case "$1" in
    --out) OUT="$2"; shift 2;;
    --limit) LIMIT="$2"; shift 2;;
    --geoms) GEOMS="$2"; shift 2;;
    *) echo "unknown argument $1" >&2; exit 2;;
  esac
done
mkdir -p "$OUT"
echo "== 1/3 generating fixtures into $OUT"
python3 tools/gen_golden.py --out "$OUT" --limit "$LIMIT" --geoms "$GEOMS" || exit 1
echo "== 2/3 which semantics do the fixtures demand?"
python3 tests/pin_semantics.py "$OUT" | tee "$OUT/pin_report.json"
PIN=${PIPESTATUS[0]}
echo "== 3/3 oracle + generator against the fixtures under the product's default semantics"
PGX_GOLDEN_DIR="$(cd "$OUT" && pwd)" python3 -m pytest tests/test_golden_reference.py -q -m "not gpu" -p no:cacheprovider
T=$?
echo "== pin_semantics rc=$PIN, default-semantics tests rc=$T (report: $OUT/pin_report.json)"
[ "$PIN" -eq 0 ] && [ "$T" -eq 0 ]
