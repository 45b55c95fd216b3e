#!/bin/bash
# round 3, session 23: the growth zoo (fishing-v5 .. v11) re-timed on the round-3 kernels, N = 2^22
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s23"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 300 python3 scripts/exp/time_zoo_lean.py > "$O/zoo_lean.jsonl" 2> "$O/err1.txt" || exit 2
timeout -k 10 300 python3 scripts/exp/time_v11.py > "$O/v11.jsonl" 2> "$O/err2.txt" || exit 3
echo done
