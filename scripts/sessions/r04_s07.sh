#!/bin/bash
# round 4, session 7: the float32 zoo as the all-float32 algebraic form on the hardware exp (math4; math4r = fishing-v11 regrouped)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s07"; mkdir -p "$O"
cd "$REPO"
for v in math4 math4r; do
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$v.so" timeout -k 10 300 python3 -m pytest tests/test_gpu_zoo.py -m gpu -x -q > "$O/pytest_$v.log" 2>&1 || { tail -30 "$O/pytest_$v.log"; echo "pytest $v failed"; }
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$v.so" timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag "$v" >> "$O/zoo_f32_error.jsonl" 2> "$O/err_$v.log" || { echo "measure $v failed"; tail -5 "$O/err_$v.log"; }
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$v.so" timeout -k 10 400 python3 scripts/exp/time_v11.py > "$O/time_v11_$v.jsonl" 2> "$O/err_v11_$v.log" || { echo "time_v11 $v failed"; }
done
echo done
