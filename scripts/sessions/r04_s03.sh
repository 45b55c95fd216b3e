#!/bin/bash
# round 4, session 3: fishing-v11 after the unpadded regroup window, msun-style float64 log / exp, exact float64 instantiations
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s03"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 500 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
timeout -k 10 400 python3 tests/measure_zoo_f32_error.py --tag default >> "$O/zoo_f32_error.jsonl" 2> "$O/err_default.log" || { echo "measure default failed"; tail -5 "$O/err_default.log"; }
echo done
