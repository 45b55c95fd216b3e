#!/bin/bash
# round 3, session 22: XCD-aware zig-zag in the product (one-tile forms and catch-alls from 100 MB per step; exact tile-loop
# twins from 500 MB [x100] or 150 MB [x100z150] per step) against the forward walk [off], fishing_step_f32, N = 2^20 .. 2^26
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s22"; mkdir -p "$O"; rm -f "$O"/ev_*
export HARNESS_SHAPE=256x4
for rnd in 1 2; do for v in z150 z150nta z150nta400; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 80 20 26 product > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
