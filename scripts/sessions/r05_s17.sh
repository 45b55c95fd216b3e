#!/bin/bash
# round 5, session 17: the catch-all kernels (noise mode at run time) waited for EVERY load of the tile before their noise generator's
# first instruction: the caller's normals (external-noise mode) were loaded with the tile's other loads into the registers the
# generator writes, so the path that never issued that load still carried its s_waitcnt vmcnt(0).  The load now sits in the
# generator's else.  zin = the product's source, zearly = -DFISHING_X_ZEXT_IN_NOISE_BRANCH=0 (before)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s17"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_zin.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_zoo.py tests/test_gpu_v4_params.py tests/test_gpu_fused_and_dispatch.py tests/test_gpu_envs.py tests/test_gpu_bounds.py -m gpu -q -x > "$O/tests_zin.log" 2>&1 || { tail -30 "$O/tests_zin.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_zin.log"
: > "$O/catch_alls.jsonl"
for rep in 1 2; do
  for var in zearly zin; do
    lib="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so"
    FISHING_HIP_LIB="$lib" timeout -k 10 400 python3 scripts/exp/time_step_sizes.py "" 22,24 2> "$O/t.err" | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); r.update(variant='$var', rep=$rep); print(json.dumps(r))" >> "$O/catch_alls.jsonl" || { echo "$var sizes failed"; tail -5 "$O/t.err"; exit 2; }
    for spec in "v4t:21:--config v4 --v4-stamped" "v4t:24:--config v4 --v4-stamped --n-envs 16777216"; do
      cfg="${spec%%:*}"; rest="${spec#*:}"; ln="${rest%%:*}"; extra="${rest#*:}"
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py $extra --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/b.err") || { echo "$var $cfg failed"; tail -5 "$O/b.err"; exit 2; }
      python3 - "$var" "$rep" "$cfg" "$ln" "$line" >> "$O/catch_alls.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[5]); r = d["roofline"]
print(json.dumps(dict(variant=sys.argv[1], rep=int(sys.argv[2]), case=sys.argv[3], log2_n=int(sys.argv[4]), kernel=r["kernel"], us=round(r["avg_launch_us"], 3))))
PY
    done
  done
done
cat "$O/catch_alls.jsonl"
