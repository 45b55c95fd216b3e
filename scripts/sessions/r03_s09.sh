#!/bin/bash
# round 3, session 9: the driver's bench command + rocprofv3 passes of the BASELINE configs after the one-tile kernels
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s09"; mkdir -p "$O"
cd "$REPO"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err" || exit 1
for spec in "v1:" "v1_2p20:--n-envs 1048576" "v2_2p19:--config v2 --n-envs 524288" "v1_2p21:--n-envs 2097152" "v0:--config v0" "v2:--config v2" "v4_21:--config v4" "v1_bare:--no-returns"; do
  tag="${spec%%:*}"; extra="${spec#*:}"
  bash scripts/profile_bench.sh "r03_s09/prof_$tag" $extra || { echo "profile $tag failed"; exit 2; }
  echo "profiled $tag"
done
# keep only what the summaries need
find "$O" -name "*agent_info.csv" -delete
echo done
