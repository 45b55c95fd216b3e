#!/bin/bash
# round 3, session 26: the requests WITHOUT a one-tile form (float64, catch-alls) on a grid of one workgroup per tile
# [fullgrid: -DFISHING_FULL_GRID_LOOPS=1] against the capped tile loop [base], N = 2^22 .. 2^26
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s26"; mkdir -p "$O"; rm -f "$O"/*.jsonl
cd "$REPO"
for rnd in 1 2; do for v in base fullgrid; do
  FISHING_HIP_LIB="$REPO/scripts/exp/_build/libs/$v/libfishing_hip.so" timeout -k 10 400 python3 scripts/exp/time_step_sizes.py >> "$O/$v.jsonl" 2> "$O/err_$v.txt" || exit 2
done; done
echo done
