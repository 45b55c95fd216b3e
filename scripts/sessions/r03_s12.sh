#!/bin/bash
# round 3, session 12: the driver's K = 20 region with the host waiting by polling (HSA_ENABLE_INTERRUPT=0) vs interrupts
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s12"; mkdir -p "$O"
cd "$REPO"
for rnd in 1 2 3; do
  for v in default poll; do
    if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-subrecords --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'variant':'$v','ms_per_step':d['ms_per_step'],'value':d['value'],'avg_launch_us':d['roofline']['avg_launch_us'],'graph':d.get('graph_region')}))" >> "$O/k20.jsonl" || exit 1
  done
done
cat "$O/k20.jsonl"
