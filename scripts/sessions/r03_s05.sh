#!/bin/bash
# round 3, session 5 (-fno-slp-vectorize: the tile loads stay ahead of the Philox block): build-variant A/B of the product's lean step kernel at N = 2^19..2^22 (trace durations + events),
# the harness's stripped step / copy of the same shape as in-process references
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s05"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HARNESS_SHAPE=256x4
for rnd in 1 2; do
for v in base noslp noslp1 noslp3 noslpall noslpf3; do
  B="$REPO/scripts/exp/_build/small_n_$v"
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_${v}_$rnd" -- "$B" 200 19 22 product,copy,step,steprec > "$OUT/ev_${v}_$rnd.jsonl" 2> "$OUT/trace.err" || exit 2
  python3 "$REPO/scripts/exp/small_n_trace.py" "$OUT/trace_${v}_$rnd" > "$OUT/trace_${v}_$rnd.jsonl" || exit 3
  rm -rf "$OUT/trace_${v}_$rnd"
done
done
echo done
