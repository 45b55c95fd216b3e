#!/bin/bash
# round 3, session 21: after xor3 (v_bitop3_b32) in the Philox rounds: tests, fishing-v4 / v1 step times, fused kernels
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s21"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -2 "$O/tests.log"
python scripts/exp/time_occupancy_cap.py > "$O/v4_v1.jsonl" 2>/dev/null
python scripts/exp/time_rollout_policies.py 2>/dev/null | tail -1 > "$O/rollout.jsonl"
python scripts/exp/time_fused.py 2>/dev/null | tail -1 > "$O/fused.jsonl"
cat "$O/v4_v1.jsonl" "$O/rollout.jsonl"
echo done
