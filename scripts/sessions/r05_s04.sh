#!/bin/bash
# round 5, session 4: the float64 zoo's hot requests as exact two-envs-per-thread instantiations, on the algebraic form
# (step times product / exact / product / exact, every zoo id in both layouts)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s04"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants"
for rep in 1 2; do
  for var in product ${VARIANTS:-f64_exact}; do
    lib="$V/libfishing_hip_$var.so"; [ "$var" = product ] && lib="$REPO/gym_fishing_amd/_lib/libfishing_hip.so"
    FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --zoo-only > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var failed"; tail -5 "$O/rows.err"; exit 2; }
  done
  echo "rep $rep done"
done
