#!/bin/bash
# round 3, session 20: GPU tests + fishing_step_f32 at N = 2^22 .. 2^26 after the nontemporal action loads (zig-zag forms, >= 800 MB per step)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r03_s20"; mkdir -p "$O"; rm -f "$O"/ev_*
cd "$REPO"
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -2 "$O/tests.log"
export HARNESS_SHAPE=256x4
for rnd in 1 2; do
  timeout -k 10 200 "$REPO/scripts/exp/_build/small_n_shapes" 60 22 26 product > "$O/ev_$rnd.jsonl" 2> "$O/err.txt" || exit 2
done
echo done
