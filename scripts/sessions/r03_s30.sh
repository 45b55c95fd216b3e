#!/bin/bash
# round 3, session 30: one-tile forms of the catch-alls [caone] against their tile loop on the same grid [product = the
# build before], float32 with terminal observations and float64, N = 2^19 .. 2^26
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s30"; mkdir -p "$O"; rm -f "$O"/*.jsonl
cd "$REPO"
for rnd in 1 2; do for v in product caone; do
  FISHING_HIP_LIB="$REPO/scripts/exp/_build/libs/$v/libfishing_hip.so" timeout -k 10 400 python3 scripts/exp/time_step_sizes.py f32_v1_term_ret,f64_v1_ret 19,20,21,22,23,24,25 >> "$O/$v.jsonl" 2> "$O/err_$v.txt" || exit 2
done; done
echo done
