#!/bin/bash
# round 3, session 17: half of the workgroups start late (s_sleep stagger) in the one-round grids: harness kernels, preload build
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s17"; mkdir -p "$O"; rm -f "$O"/*.jsonl
export HARNESS_SHAPE=256x4
for mode in 0 1 2; do for d in 0 4 8 12 16 24 32 48; do
  [ $mode != 0 ] && [ $d = 0 ] && continue
  HARNESS_STAGGER=$d HARNESS_STAGGER_MODE=$mode timeout -k 10 120 "$REPO/scripts/exp/_build/small_n_stagger" 300 18 22 copy,step,steprec | sed "s/^{/{\"stagger\": $d, \"mode\": $mode, /" >> "$O/stagger.jsonl" || exit 2
done; done
echo done
