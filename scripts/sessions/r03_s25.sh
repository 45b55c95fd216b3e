#!/bin/bash
# round 3, session 25: one tile per workgroup at N = 2^25 / 2^26 (32768 / 65536 return_partials slots) against the capped loop
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s25"; mkdir -p "$O"; rm -f "$O"/ev_*
export HARNESS_SHAPE=256x4
for rnd in 1 2; do for v in s4k s32k s64k; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 80 25 26 product > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
