#!/bin/bash
# round 4, session 4: fishing-v11 float32 as the select form (no regroup) + the whole zoo re-timed; rollouts / fused steps of v11
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s04"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 500 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
timeout -k 10 400 python3 tests/measure_zoo_f32_error.py --tag default >> "$O/zoo_f32_error.jsonl" 2> "$O/err_default.log" || { echo "measure default failed"; tail -5 "$O/err_default.log"; }
timeout -k 10 400 python3 scripts/exp/time_v11.py > "$O/time_v11.jsonl" 2> "$O/err_v11.log" || { echo "time_v11 failed"; tail -5 "$O/err_v11.log"; }
echo done
