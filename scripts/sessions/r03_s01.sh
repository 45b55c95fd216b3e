#!/bin/bash
# round 3, session 1: small-N shape sweep (C-enqueued copy ceiling vs step bodies), events + rocprofv3 durations
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s01"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/scripts/exp/_build/small_n_shapes"
timeout -k 10 240 "$B" 400 17 21 > "$OUT/events.jsonl" 2> "$OUT/events.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- "$B" 100 17 21 > "$OUT/events_under_prof.jsonl" 2> "$OUT/trace.err" || exit 2
python3 "$REPO/scripts/exp/small_n_trace.py" "$OUT/trace" > "$OUT/trace.jsonl" || exit 3
rm -rf "$OUT/trace"
echo done
