#!/bin/bash
# round 5, session 14: fishing-v4's year counters loaded FIRST in the derived-parameter exact kernels (the (K, r) derivation is the one
# piece of arithmetic that needs loaded data before it can start; loads return in the order they were issued) and the sigma array's
# load last.  tfirst = the product's source, tthird = -DFISHING_X_V4_T_FIRST=0 (before: sigma array, observations, year counters)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s14"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_tfirst.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_v4_params.py tests/test_gpu_fused_and_dispatch.py tests/test_gpu_envs.py -m gpu -q -x > "$O/tests_tfirst.log" 2>&1 || { tail -30 "$O/tests_tfirst.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_tfirst.log"
: > "$O/tfirst.jsonl"
for rep in 1 2 3; do
  for var in tthird tfirst; do
    lib="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so"
    for spec in "v4:21:--config v4" "v4:22:--config v4" "v4:24:--config v4" "v4:20:--config v4"; do
      cfg="${spec%%:*}"; rest="${spec#*:}"; ln="${rest%%:*}"; extra="${rest#*:}"
      n=$((1 << ln))
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py $extra --n-envs $n --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/b.err") || { echo "$var $cfg $ln failed"; tail -5 "$O/b.err"; exit 2; }
      python3 - "$var" "$rep" "$cfg" "$ln" "$line" >> "$O/tfirst.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[5]); r = d["roofline"]
print(json.dumps(dict(variant=sys.argv[1], rep=int(sys.argv[2]), config=sys.argv[3], log2_n=int(sys.argv[4]), kernel=r["kernel"],
                      avg_launch_us=round(r["avg_launch_us"], 3), frac=r["frac"], hbm_spec_ratio=r.get("hbm_spec_ratio"))))
PY
    done
  done
done
cat "$O/tfirst.jsonl"
