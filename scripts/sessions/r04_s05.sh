#!/bin/bash
# round 4, session 5: the float32 zoo in the hybrid form (default) against the float64 form (math2) and the hardware round trip (math0)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s05"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 500 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
timeout -k 10 400 python3 tests/measure_zoo_f32_error.py --tag hybrid_default >> "$O/zoo_f32_error.jsonl" 2> "$O/err_default.log" || { echo "measure default failed"; tail -5 "$O/err_default.log"; }
timeout -k 10 400 python3 scripts/exp/time_v11.py > "$O/time_v11.jsonl" 2> "$O/err_v11.log" || { echo "time_v11 failed"; tail -5 "$O/err_v11.log"; }
for v in math0 math2; do
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$v.so" timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag "$v" >> "$O/zoo_f32_error.jsonl" 2> "$O/err_$v.log" || { echo "measure $v failed"; tail -5 "$O/err_$v.log"; }
done
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_math0.so" timeout -k 10 400 python3 scripts/exp/time_v11.py > "$O/time_v11_math0.jsonl" 2> "$O/err_v11_math0.log" || { echo "time_v11 math0 failed"; }
echo done
