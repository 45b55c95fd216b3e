#!/bin/bash
# round 3, session 7: raw per-dispatch timeline at N = 2^19 (what is the bimodal launch duration?)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03_s07"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HARNESS_SHAPE=256x4
B="$REPO/scripts/exp/_build/small_n_base"
HARNESS_RR=1 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_rr" -- "$B" 1500 19 19 product,copy,step,steprec > "$OUT/ev_rr.jsonl" 2> "$OUT/trace.err" || exit 2
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_seq" -- "$B" 300 19 19 product,copy,step,steprec > "$OUT/ev_seq.jsonl" 2> "$OUT/trace.err" || exit 2
find "$OUT" -name "*kernel_trace.csv" -exec sh -c 'cut -d, -f1-20 "$1" > "$1.cut"' _ {} \;
find "$OUT" -name "*agent_info.csv" -delete
rocm-smi --showclocks > "$OUT/clocks.txt" 2>&1
echo done
