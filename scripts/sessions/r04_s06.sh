#!/bin/bash
# round 4, session 6: fishing-v11 float32 hybrid bodies, select form (default build of session 5) vs regrouped through LDS
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s06"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_regroup.so" timeout -k 10 300 python3 -m pytest tests/test_gpu_zoo.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_regroup.so" timeout -k 10 400 python3 scripts/exp/time_v11.py > "$O/time_v11_regroup.jsonl" 2> "$O/err_v11.log" || { echo "time_v11 failed"; tail -5 "$O/err_v11.log"; }
echo done
