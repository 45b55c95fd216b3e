#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 SQ-counter pass over scripts/exp/time_fused.py (fused K-step kernel vs per-step
# launches at N = 2^18 .. 2^22, fishing-v1 / v2 / v4).  scripts/summarize_sq.py reduces it.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-sq_fused}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_sq" -- \
    python3 "$REPO/scripts/exp/time_fused.py" > "$OUT/time_fused.json" 2> "$OUT/sq.err" || exit 1
echo "sq pass done: $OUT"
