#!/bin/bash
# One round's judged records, on the GPU box (through gpurun): the driver bench command, the default line, and the
# rocprofv3 --kernel-trace --stats + two --pmc passes (scripts/profile_bench.sh) of every BASELINE config, SURVEY 8(d)'s
# spill sizes and the float64 layout, each reduced by scripts/summarize_profile.py into gpurun_out/<tag>/summ/
# (copy what is to be judged into profiles/).
#   bash scripts/profile_round.sh r06 A      # bench lines + the BASELINE configs' cache-resident sizes
#   bash scripts/profile_round.sh r06 B      # config 5 whole, the spill sizes of configs 3 / 4, the float64 metric, N = 2^21
#   bash scripts/profile_round.sh r06 C      # fishing-v4's other modes, the bare step, N = 2^24 / 2^26, float64 at 2^24
# (three gpurun calls: each stays inside one call's time limit)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r06}"
PART="${2:-A}"
O="$REPO/gpurun_out/${TAG}_prof"; mkdir -p "$O/summ"
cd "$REPO"
# which box, and its clocks / temperatures / memory state before and after (the float64 N = 2^24 step has read 129 us on some boxes and
# 138 us on others with the same library: whatever differs between boxes should show here)
smi() { { hostname; date +%s; rocm-smi --showclocks --showtemp --showpower --showmemuse --showperflevel --json 2>/dev/null; } > "$O/smi_${PART}_$1.txt" 2>&1 || true; }
smi before
if [ "$PART" = A ]; then
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err" || exit 1
  echo "driver line done"
  timeout -k 10 400 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
  echo "default line done"
  SPECS=("v1:" "v1_2p20:--n-envs 1048576" "v2_2p19:--config v2 --n-envs 524288" "v0:--config v0" "v2:--config v2" "v4_21:--config v4")
elif [ "$PART" = B ]; then
  SPECS=("v4_24:--config v4 --n-envs 16777216" "v0_2p26:--config v0 --n-envs 67108864" "v2_2p26:--config v2 --n-envs 67108864" "v1_f64:--f64" "v1_f64_bare:--f64 --no-returns" "v1_2p21:--n-envs 2097152")
else
  SPECS=("v4s_21:--config v4 --v4-stored" "v4t_21:--config v4 --v4-stamped" "v1_bare:--no-returns" "v1_2p24:--n-envs 16777216" "v1_2p26:--n-envs 67108864" "v1_f64_2p24:--f64 --n-envs 16777216")
fi
for spec in "${SPECS[@]}"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_bench.sh "${TAG}_prof/prof_$tag" $extra > /dev/null || { echo "profile $tag failed"; exit 2; }
  python3 scripts/summarize_profile.py "$O/prof_$tag" "$O/summ/${TAG}_step_$tag" --latest > /dev/null || { echo "summary $tag failed"; exit 3; }
  rm -rf "$O/prof_$tag"
  echo "profiled $tag"
  case "$tag" in v1_f64_2p24|v1_2p26|v4_24) smi "after_$tag";; esac
done
smi after
echo done
