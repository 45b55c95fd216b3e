#!/bin/bash
# round 5, session 16: fishing-v11's index load behind the loads that need nothing but preloaded arguments (its pointer is a field of
# the by-value struct: in front, it made every load of the tile wait for the s_load batch).  klate = the product's source,
# kfirst = -DFISHING_X_V11_KINDS_LATE=0
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s16"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_klate.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_fused_and_dispatch.py -m gpu -q -x > "$O/tests_klate.log" 2>&1 || { tail -30 "$O/tests_klate.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_klate.log"
for rep in 1 2 3; do
  for var in kfirst klate; do
    FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --v11-only > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
    FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 300 python3 scripts/exp/time_v11.py > "$O/sizes_${var}_$rep.jsonl" 2>> "$O/rows.err" || { echo "$var sizes failed"; tail -5 "$O/rows.err"; exit 3; }
  done
done
echo done
