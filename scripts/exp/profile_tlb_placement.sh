#!/bin/bash
#   bash scripts/exp/profile_tlb_placement.sh [log2_n] [trials]
# Run ON THE GPU BOX: UTCL1 (per-CU address-translation cache) counters per step-kernel dispatch while
# scripts/exp/placement_large.py re-allocates the env arena each trial (mode 2): do the slow allocations miss more?
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
LN="${1:-26}"
TRIALS="${2:-8}"
OUT="$REPO/gpurun_out/stag/pmc_tlb"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum \
    --output-format csv -d "$OUT" -- \
    python3 "$REPO/scripts/exp/placement_large.py" "$LN" "$TRIALS" 0 2 > "$OUT/trials.jsonl" 2> "$OUT/err.txt" || exit 1
echo "tlb pass done: $OUT"
