#!/bin/bash
# round 5, session 12: fishing-v11's indices clamped into the zoo where they are loaded (one copy in registers instead of two: 65 -> 62
# VGPRs in the float32 kernel with returns = eight waves per SIMD instead of seven).  clamp = the product's source, noclamp = before
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s12"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_clamp.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_fused_and_dispatch.py tests/test_gpu_envs.py -m gpu -q -x > "$O/tests_clamp.log" 2>&1 || { tail -30 "$O/tests_clamp.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_clamp.log"
for rep in 1 2 3; do
  for var in noclamp clamp; do
    FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --v11-only > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
    FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 300 python3 scripts/exp/time_v11.py > "$O/sizes_${var}_$rep.jsonl" 2>> "$O/rows.err" || { echo "$var sizes failed"; tail -5 "$O/rows.err"; exit 3; }
  done
done
echo done
