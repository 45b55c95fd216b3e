#!/bin/bash
# round 5, session 15: the year counters' load behind the actions' in the exact kernels other than fishing-v4's derived ones
# (-DFISHING_X_T_LATE=1: the arithmetic starts on observations and actions -- first wait vmcnt(2) instead of vmcnt(1) of four loads).
# base = the product's source, tlate = the experiment
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s15"; mkdir -p "$O"
cd "$REPO"
FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_tlate.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_zoo.py tests/test_gpu_fused_and_dispatch.py -m gpu -q -x > "$O/tests_tlate.log" 2>&1 || { tail -30 "$O/tests_tlate.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_tlate.log"
: > "$O/tlate.jsonl"
for rep in 1 2 3; do
  for var in base tlate; do
    lib="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so"
    for spec in "v1:22:" "v1:21:" "v1:20:" "v2:19:--config v2" "v0:22:--config v0" "v1f64:22:--f64" "v1:24:"; do
      cfg="${spec%%:*}"; rest="${spec#*:}"; ln="${rest%%:*}"; extra="${rest#*:}"
      n=$((1 << ln))
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py $extra --n-envs $n --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/b.err") || { echo "$var $cfg $ln failed"; tail -5 "$O/b.err"; exit 2; }
      python3 - "$var" "$rep" "$cfg" "$ln" "$line" >> "$O/tlate.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[5]); r = d["roofline"]
print(json.dumps(dict(variant=sys.argv[1], rep=int(sys.argv[2]), config=sys.argv[3], log2_n=int(sys.argv[4]), kernel=r["kernel"],
                      avg_launch_us=round(r["avg_launch_us"], 3), frac=r["frac"], hbm_spec_ratio=r.get("hbm_spec_ratio"))))
PY
    done
    [ $rep = 3 ] || FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --zoo-only > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
  done
done
cat "$O/tlate.jsonl"
