#!/bin/bash
# round 5, session 1: the float64 zoo on the algebraic form (variant library) -- zoo tests, then step times old / new / old / new
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s01"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants/libfishing_r05a.so"
FISHING_HIP_LIB="$V" timeout -k 10 600 python3 -m pytest tests/test_gpu_zoo.py -m gpu -x -q > "$O/zoo_tests_variant.log" 2>&1 || { tail -30 "$O/zoo_tests_variant.log"; exit 1; }
tail -2 "$O/zoo_tests_variant.log"
for rep in 1 2; do
  timeout -k 10 300 python3 scripts/exp/run_f_rows.py --zoo-only > "$O/rows_base_$rep.jsonl" 2> "$O/rows_base_$rep.err" || exit 2
  FISHING_HIP_LIB="$V" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --zoo-only > "$O/rows_variant_$rep.jsonl" 2> "$O/rows_variant_$rep.err" || exit 3
  echo "rep $rep done"
done
