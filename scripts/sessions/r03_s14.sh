#!/bin/bash
# round 3, session 14: kernarg preload (-mllvm -amdgpu-kernarg-preload-count=N, stream pointers as leading scalar arguments)
# in the harness: N = 4 / 8 / 10 / 14 preloaded dwords vs none, back-to-back HIP events, N = 2^18 .. 2^22
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s14"; mkdir -p "$O"
export HARNESS_SHAPE=256x4
for rnd in 1 2 3; do for v in scalar pre4 pre8 pre10 preload; do
  timeout -k 10 120 "$REPO/scripts/exp/_build/small_n_kp_$v" 300 18 22 copy,step,steprec > "$O/ev_${v}_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done; done
echo done
