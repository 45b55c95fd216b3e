#!/bin/bash
# round 3, session 15: the product's lean kernel with / without kernarg preload and with / without the argument batch in
# the one-tile forms (variant libraries selected through LD_LIBRARY_PATH), back-to-back HIP events
#   C no preload, batch up front (round 3 before this)   D preload, batch up front   E preload, batch behind the tile loads in the one-tile forms
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s15"; mkdir -p "$O"; rm -f "$O"/ev_*
export HARNESS_SHAPE=256x4
for rnd in 1 2 3; do for v in C D E; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 120 "$REPO/scripts/exp/_build/small_n_shapes" 300 18 22 product,copy > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
