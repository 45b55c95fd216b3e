#!/bin/bash
# round 3, session 28: with a workgroup per tile at every size -- (a) the product's nontemporal action loads (from 200 MB
# per step) and its XCD-aware zig-zag (from 100 MB) each switched off, N = 2^23 .. 2^26; (b) the same-shape copy walking
# forward and walking like the step (HARNESS_STAGGER_MODE=7)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s28"; mkdir -p "$O"; rm -f "$O"/*.jsonl
export HARNESS_SHAPE=256x4
for rnd in 1 2; do for v in product nta_off zz_off; do
  LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/$v" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 60 23 26 product > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/product" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 60 23 26 copy > "$O/copy_forward.jsonl" 2>> "$O/err.txt" || exit 3
HARNESS_STAGGER=1 HARNESS_STAGGER_MODE=7 LD_LIBRARY_PATH="$REPO/scripts/exp/_build/libs/product" timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 60 23 26 copy > "$O/copy_xzz.jsonl" 2>> "$O/err.txt" || exit 4
echo done
