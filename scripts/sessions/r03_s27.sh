#!/bin/bash
# round 3, session 27: beyond the return_partials slots (N = 2^27 = 131072 tiles): the tile loop on 768 / 4096 / 65536 workgroups
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s27"; mkdir -p "$O"; rm -f "$O"/ev_*
export HARNESS_SHAPE=256x4
for rnd in 1 2; do for v in cap768 cap4k cap64k; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 40 27 27 product > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
