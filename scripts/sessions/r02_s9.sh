#!/bin/bash
set -u
O=gpurun_out/r02_s9
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for r in 0 1; do for v in base rollkeys0; do echo "== rollout $v"; FISHING_HIP_LIB=gym_fishing_amd/_lib/variants/libfishing_hip_$v.so timeout -k 10 300 python scripts/exp/time_rollout_policies.py 2>&1 | tail -1 | tee -a $O/rollout_$v.jsonl; done; done
echo "== A/B base"; timeout -k 10 600 python scripts/exp/ab_r02.py base > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl
