#!/bin/bash
# round 5, session 3: fishing-v11 on the select form (one division + one exp per env, no regroup) and the Philox2x32 model redraw --
# full GPU suite on the product, then step / rollout times: round-4 equivalent (v11_base) / this tree with the 4x32 redraw / product (2x32 redraw, regroup) / select form
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s03"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants"
timeout -k 10 600 python3 -m pytest tests -m gpu -q > "$O/gpu_tests.log" 2>&1 || { tail -40 "$O/gpu_tests.log" | cut -c1-250; exit 1; }
tail -1 "$O/gpu_tests.log"
: > "$O/v11_forms.jsonl"
for rep in 1 2; do
  for var in v11_base v11_redraw4x32 product v11_select; do
    lib="$V/libfishing_hip_$var.so"; [ "$var" = product ] && lib="$REPO/gym_fishing_amd/_lib/libfishing_hip.so"
    FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --v11-only > "$O/rows.tmp" 2> "$O/rows.err" || { echo "$var failed"; tail -5 "$O/rows.err"; exit 2; }
    python3 - "$var" "$rep" "$O/rows.tmp" >> "$O/v11_forms.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[3]):
    d = json.loads(line); d.update(variant=sys.argv[1], rep=int(sys.argv[2])); print(json.dumps(d))
PY
  done
  echo "rep $rep done"
done
echo done
