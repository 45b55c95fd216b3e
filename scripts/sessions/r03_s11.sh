#!/bin/bash
# round 3, session 11: the copy ceiling of the product's stream shape at HBM-resident sizes (one tile per workgroup, no walk order)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s11"; mkdir -p "$O"
B="$REPO/scripts/exp/_build/small_n_shapes"
HARNESS_SHAPE=256x4 timeout -k 10 300 "$B" 60 24 26 product,copy,step > "$O/large_256x4.jsonl" 2> "$O/err.txt" || exit 1
HARNESS_SHAPE=512x4 timeout -k 10 300 "$B" 60 24 26 copy,step > "$O/large_512x4.jsonl" 2>> "$O/err.txt" || exit 1
HARNESS_SHAPE=256x8 timeout -k 10 300 "$B" 60 24 26 copy,step > "$O/large_256x8.jsonl" 2>> "$O/err.txt" || exit 1
echo done
