#!/bin/bash
# round 5, session 10: the driver bench command + rocprofv3 / PMC passes of every BASELINE config (and SURVEY 8(d)'s spill sizes of
# configs 3 / 4) on the round's final kernels (= scripts/sessions/r04_s10.sh + v0 / v2 at 2^26)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s10"; mkdir -p "$O"
cd "$REPO"
# (two gpurun calls: `bash scripts/sessions/r05_s10.sh A` = the bench lines + the BASELINE configs, `... B` = the other sizes and layouts)
PART="${1:-A}"
if [ "$PART" = A ]; then
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err" || exit 1
  echo "driver line done"
  timeout -k 10 400 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
  echo "default line done"
  SPECS=("v1:" "v1_2p20:--n-envs 1048576" "v2_2p19:--config v2 --n-envs 524288" "v0:--config v0" "v2:--config v2" "v4_21:--config v4" "v4_24:--config v4 --n-envs 16777216" "v0_2p26:--config v0 --n-envs 67108864" "v2_2p26:--config v2 --n-envs 67108864")
else
  SPECS=("v1_2p21:--n-envs 2097152" "v4s_21:--config v4 --v4-stored" "v4t_21:--config v4 --v4-stamped" "v1_bare:--no-returns" "v1_2p24:--n-envs 16777216" "v1_2p26:--n-envs 67108864" "v1_f64_2p24:--f64 --n-envs 16777216" "v1_f64_bare:--f64 --no-returns" "v1_f64:--f64")
fi
for spec in "${SPECS[@]}"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_bench.sh "r05_s10/prof_$tag" $extra > /dev/null || { echo "profile $tag failed"; exit 2; }
  mkdir -p "$O/summ"
  python3 scripts/summarize_profile.py "$O/prof_$tag" "$O/summ/r05_step_$tag" --latest > /dev/null || { echo "summary $tag failed"; exit 3; }
  rm -rf "$O/prof_$tag"
  echo "profiled $tag"
done
echo done
