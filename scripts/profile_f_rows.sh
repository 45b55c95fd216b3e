#!/bin/bash
#   bash scripts/profile_f_rows.sh <tag>
# Run ON THE GPU BOX: rocprofv3 over scripts/exp/run_f_rows.py (the SURVEY 8(f) kernels: zoo step kernels, fused K-step
# kernel, in-kernel-policy rollouts) -- kernel-trace + stats, then the two PMC passes (separate: TCC has 4 slots) and one
# SQ pass.  Raw CSVs under gpurun_out/<tag>/; scripts/summarize_f_rows.py reduces them to profiles/.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-f_rows}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$REPO/scripts/exp/run_f_rows.py" > "$OUT/rows.jsonl" 2> "$OUT/trace.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 "$REPO/scripts/exp/run_f_rows.py" --quick > /dev/null 2> "$OUT/pmc_fetch.err" || exit 2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 "$REPO/scripts/exp/run_f_rows.py" --quick > /dev/null 2> "$OUT/pmc_write.err" || exit 3
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_sq" -- \
    python3 "$REPO/scripts/exp/run_f_rows.py" --quick > /dev/null 2> "$OUT/pmc_sq.err" || exit 4
echo "f-row passes done: $OUT"
