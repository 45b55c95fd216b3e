#!/bin/bash
# round 5, session 11: the launch's walk (zig-zag, its parity, nontemporal action loads) in the PRELOADED n_live argument instead of
# the by-value struct -- as struct fields they put an s_load round trip in front of every wave's first global load (found in the ISA:
# s_waitcnt lgkmcnt(0) at instruction 8, the first global_load at 55 of fishing::step_kernel_lean<float, 1, 12294, 4>).
# Variants (scripts/build_variants.py): struct = -DFISHING_WALK_PRELOADED=0 (rounds 4 - 5a), walk = the product's source,
# nodev = -DFISHING_X_WALK_NO_DEVICE (no branch for a device-held step counter: the bound of what the walk word can give)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s11"; mkdir -p "$O"
cd "$REPO"
VARS="${VARS:-struct walk nodev}"
# (second call: VARS="walk luthost v4s" PART=2 -- luthost = + fishing-v11's coefficient table made on the host, v4s = + fishing-v4's
# reset origin by scalar loads behind the tile's loads; the product = v4s)
if [ "${PART:-1}" = 1 ]; then
  SPECS=("v1:22:" "v1:21:" "v1:20:" "v2:19:--config v2" "v0:22:--config v0" "v4:21:--config v4" "v4:24:--config v4" "v1f64:22:--f64" "v1:24:" "v1:26:")
else
  SPECS=("v4:21:--config v4" "v4:22:--config v4" "v4:24:--config v4" "v4s:21:--config v4 --v4-stored" "v1:20:" "v1:22:")
fi
for var in $VARS; do
  [ "$var" = struct ] && continue
  FISHING_HIP_LIB="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_zoo.py tests/test_gpu_v4_params.py tests/test_gpu_fused_and_dispatch.py tests/test_gpu_envs.py -m gpu -q -x > "$O/tests_$var.log" 2>&1 || { tail -30 "$O/tests_$var.log" | cut -c1-250; exit 1; }
  echo "$var: $(tail -1 "$O/tests_$var.log")"
done
: > "$O/walk${PART:-1}.jsonl"
for rep in 1 2; do
  for var in $VARS; do
    lib="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_$var.so"
    for spec in "${SPECS[@]}"; do
      cfg="${spec%%:*}"; rest="${spec#*:}"; ln="${rest%%:*}"; extra="${rest#*:}"
      n=$((1 << ln))
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py $extra --n-envs $n --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/b.err") || { echo "$var $cfg $ln failed"; tail -5 "$O/b.err"; exit 2; }
      python3 - "$var" "$rep" "$cfg" "$ln" "$line" >> "$O/walk${PART:-1}.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[5]); r = d["roofline"]
print(json.dumps(dict(variant=sys.argv[1], rep=int(sys.argv[2]), config=sys.argv[3], log2_n=int(sys.argv[4]), kernel=r["kernel"],
                      avg_launch_us=round(r["avg_launch_us"], 3), frac=r["frac"], hbm_spec_ratio=r.get("hbm_spec_ratio"))))
PY
    done
    FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py --v11-only > "$O/rows${PART:-1}_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
  done
  echo "rep $rep done"
done
echo done
