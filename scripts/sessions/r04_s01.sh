#!/bin/bash
# round 4, session 1: the float32 zoo's distance from the reference under three evaluations of the growth functions
# (hardware transcendentals / library logf-expf / float64 mu rounded once), the new float32 golden test of the core path,
# and SQ-counter passes of the kernels DESIGN calls VALU-bound (fishing-v4 at its config-5 shard, the fused kernels)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r04_s01"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 500 python3 -m pytest tests/test_gpu_zoo.py tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.log" 2>&1 || { tail -30 "$O/pytest.log"; echo "pytest failed"; }
for v in math0 math1; do
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$v.so" timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag "$v" >> "$O/zoo_f32_error.jsonl" 2> "$O/err_$v.log" || { echo "measure $v failed"; tail -5 "$O/err_$v.log"; }
done
timeout -k 10 300 python3 tests/measure_zoo_f32_error.py --tag math2_default >> "$O/zoo_f32_error.jsonl" 2> "$O/err_default.log" || { echo "measure default failed"; tail -5 "$O/err_default.log"; }
for spec in "v4_2p21:--config v4" "v1_2p22_extra:--extra"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_sq.sh "r04_s01/sq_$tag" $extra > /dev/null || { echo "sq $tag failed"; exit 2; }
  python3 scripts/summarize_sq.py "$O/sq_$tag" "$O/r04_sq_$tag.json" > "$O/r04_sq_$tag.txt" || exit 3
  rm -rf "$O/sq_$tag/pmc_sq"
done
bash scripts/profile_sq_fused.sh "r04_s01/sq_fused" > /dev/null || { echo "sq fused failed"; exit 2; }
python3 scripts/summarize_sq.py "$O/sq_fused" "$O/r04_sq_fused.json" > "$O/r04_sq_fused.txt" || exit 3
rm -rf "$O/sq_fused/pmc_sq"
echo done
