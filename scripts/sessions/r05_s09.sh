#!/bin/bash
# round 5, session 9: rocprofv3 records of the SURVEY 8(f) kernels (zoo steps in both layouts, fused step, rollouts) on the round's final kernels
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s09"; mkdir -p "$O"
cd "$REPO"
bash scripts/profile_f_rows.sh r05_s09/f_rows > "$O/f_rows.log" 2>&1 || { echo "f rows failed"; tail -5 "$O/f_rows.log"; tail -5 "$O/f_rows/trace.err"; exit 2; }
mkdir -p "$O/summ"
python3 scripts/summarize_f_rows.py "$O/f_rows" "$O/summ/r05" > "$O/summ.log" 2>&1 || { echo "summary failed"; tail -20 "$O/summ.log"; exit 3; }
cp "$O/f_rows/rows.jsonl" "$O/summ/r05_f_rows_events.jsonl"
rm -rf "$O/f_rows/trace" "$O/f_rows/pmc_fetch" "$O/f_rows/pmc_write" "$O/f_rows/pmc_sq"
echo done
