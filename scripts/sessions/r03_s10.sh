#!/bin/bash
# round 3, session 10: N = 2^26 step time vs how the arena was allocated (hipMalloc / per-stream / VMM API), returns + bare
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s10"; mkdir -p "$O"
B="$REPO/scripts/exp/_build/placement_vmm"
timeout -k 10 300 "$B" 26 4 1 > "$O/placement_ret.jsonl" 2> "$O/err.txt" || { tail -3 "$O/err.txt"; exit 1; }
timeout -k 10 300 "$B" 26 3 0 > "$O/placement_bare.jsonl" 2>> "$O/err.txt" || { tail -3 "$O/err.txt"; exit 1; }
timeout -k 10 300 "$B" 24 3 1 > "$O/placement_ret_24.jsonl" 2>> "$O/err.txt" || { tail -3 "$O/err.txt"; exit 1; }
echo done
