#!/bin/bash
# round 3, session 6: build-variant A/B, ROUND-ROBIN launches (every kernel sees the same clock state): trace durations
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s06"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HARNESS_SHAPE=256x4
for rnd in 1 2; do
for v in base noslp noslp1 noslp3 noslpall noslp1nl; do
  B="$REPO/scripts/exp/_build/small_n_$v"
  HARNESS_RR=$rnd timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${v}_$rnd" -- "$B" 1500 19 22 product,copy,step,steprec > "$OUT/ev_${v}_$rnd.jsonl" 2> "$OUT/trace.err" || exit 2
  python3 "$REPO/scripts/exp/small_n_trace.py" "$OUT/trace_${v}_$rnd" > "$OUT/trace_${v}_$rnd.jsonl" || exit 3
  rm -rf "$OUT/trace_${v}_$rnd"
done
done
echo done
