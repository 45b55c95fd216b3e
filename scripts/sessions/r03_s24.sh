#!/bin/bash
# round 3, session 24: one tile per workgroup beyond 4096 tiles (return_partials with 16384 / 32768 slots) against the
# capped tile loop (4096 slots), fishing_step_f32 bare and with returns, N = 2^22 .. 2^26
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s24"; mkdir -p "$O"; rm -f "$O"/ev_*
export HARNESS_SHAPE=256x4
for rnd in 1 2; do for v in s4k s16k s32k; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 80 22 26 product > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
