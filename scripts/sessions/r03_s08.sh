#!/bin/bash
# round 3, session 8: build-variant A/B by back-to-back HIP-event timing (no profiler), 3 interleaved rounds
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s08"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HARNESS_SHAPE=256x4
for rnd in 1 2 3; do
for v in base noslp noslp1 noslp3 noslpall noslp1nl; do
  B="$REPO/scripts/exp/_build/small_n_$v"
  timeout -k 10 120 "$B" 400 18 22 product,copy,step,steprec,empty > "$OUT/ev_${v}_$rnd.jsonl" 2> "$OUT/err.txt" || exit 2
done
done
echo done
