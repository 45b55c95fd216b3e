#!/bin/bash
# round 5, session 6: fishing-v11 with the growth functions' coefficients from an LDS table (-DFISHING_V11_FORM=2): zoo tests on the
# variant, then step / rollout times product / variant, alternating
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s06"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_v11_lut.so"
FISHING_HIP_LIB="$V" timeout -k 10 600 python3 -m pytest tests/test_gpu_zoo.py -m gpu -q -x > "$O/tests_variant.log" 2>&1 || { tail -30 "$O/tests_variant.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_variant.log"
: > "$O/v11_lut.jsonl"
for rep in 1 2; do
  for var in product v11_lut; do
    lib="$V"; [ "$var" = product ] && lib="$REPO/gym_fishing_amd/_lib/libfishing_hip.so"
    FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --v11-only > "$O/rows.tmp" 2> "$O/rows.err" || { echo "$var failed"; tail -5 "$O/rows.err"; exit 2; }
    python3 - "$var" "$rep" "$O/rows.tmp" >> "$O/v11_lut.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[3]):
    d = json.loads(line); d.update(variant=sys.argv[1], rep=int(sys.argv[2])); print(json.dumps(d))
PY
  done
  echo "rep $rep done"
done
echo done
