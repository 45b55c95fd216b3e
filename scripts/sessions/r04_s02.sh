#!/bin/bash
# round 4, session 2: the float32 zoo evaluated as the float64 algebraic form (default) against the hardware round trip (math0)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s02"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 500 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag alg_f64_default >> "$O/zoo_f32_error.jsonl" 2> "$O/err_default.log" || { echo "measure default failed"; tail -5 "$O/err_default.log"; }
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_math0.so" timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag math0 >> "$O/zoo_f32_error.jsonl" 2> "$O/err_math0.log" || { echo "measure math0 failed"; tail -5 "$O/err_math0.log"; }
echo done
