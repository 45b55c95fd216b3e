#!/bin/bash
#   bash scripts/profile_sq.sh <tag> [bench.py flags; default --extra]
# Run ON THE GPU BOX: one rocprofv3 SQ-counter pass (8 SQ slots on gfx950) over `bench.py --extra`, which runs
# the step kernels (HBM-bound) and the fused rollout kernels (VALU-bound).  scripts/summarize_sq.py reduces
# the per-dispatch rows to per-kernel wave-cycle shares.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-sq}"
shift || true
EXTRA="${*:---extra}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_sq" -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-subrecords $EXTRA --steps 101 --warmup 20 --spinup-ms 20 > "$OUT/bench.json" 2> "$OUT/sq.err" || exit 1
echo "sq pass done: $OUT"
