#!/bin/bash
# round 3, session 4: return-record variants incl. prefetched read-modify-write of the workgroup's slot (no atomic)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s04"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HARNESS_SHAPE=256x4
B="$REPO/scripts/exp/_build/small_n_shapes"
for rnd in 1 2 3; do
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$rnd" -- "$B" 200 18 22 product,copy,step,steprec,empty > "$OUT/ev_$rnd.jsonl" 2> "$OUT/trace.err" || exit 2
  python3 "$REPO/scripts/exp/small_n_trace.py" "$OUT/trace_$rnd" > "$OUT/trace_$rnd.jsonl" || exit 3
  rm -rf "$OUT/trace_$rnd"
done
timeout -k 10 120 "$B" 400 18 22 product,copy,step,steprec,empty > "$OUT/events.jsonl"
echo done
