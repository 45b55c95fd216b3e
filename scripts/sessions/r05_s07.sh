#!/bin/bash
# round 5, session 7: the noise generator pinned in front of the first wait for the tile's loads.  In the fishing-v4 derived kernels and in
# the Beverton-Holt / Myers / May kernels of both layouts LLVM had SUNK the generator behind the control flow that follows it (the (K, r)
# derivation; the growth function's power / quotient branches) -- the wave waited for its loads before doing any arithmetic.  A scheduling
# fence behind the noise block (-DFISHING_LEAN_FENCE=3, first run of this session) does not bind an IR-level move; an empty asm that
# reads the normals does (FISHING_LEAN_PIN_NOISE, the product).  VARIANT=nopin = the library before
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s07"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_${VARIANT:-nopin}.so"
FISHING_HIP_LIB="$V" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_zoo.py tests/test_gpu_v4_params.py tests/test_gpu_fused_and_dispatch.py -m gpu -q -x > "$O/tests_variant.log" 2>&1 || { tail -30 "$O/tests_variant.log" | cut -c1-250; exit 1; }
tail -1 "$O/tests_variant.log"
: > "$O/fence.jsonl"
for rep in 1 2; do
  for var in product variant pinall; do
    lib="$V"; [ "$var" = product ] && lib="$REPO/gym_fishing_amd/_lib/libfishing_hip.so"; [ "$var" = pinall ] && lib="$REPO/gym_fishing_amd/_lib/variants/libfishing_hip_pinall.so"
    for spec in "v1:22:" "v1:21:" "v1:20:" "v2:19:--config v2" "v2:22:--config v2" "v0:22:--config v0" "v4:21:--config v4" "v4:22:--config v4" "v4:24:--config v4" "v4s:21:--config v4 --v4-stored" "v1f64:22:--f64" "v1:26:"; do
      cfg="${spec%%:*}"; rest="${spec#*:}"; ln="${rest%%:*}"; extra="${rest#*:}"
      n=$((1 << ln))
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py $extra --n-envs $n --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/b.err") || { echo "$var $cfg $ln failed"; tail -5 "$O/b.err"; exit 2; }
      python3 - "$var" "$rep" "$cfg" "$ln" "$line" >> "$O/fence.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[5]); r = d["roofline"]
print(json.dumps(dict(variant=sys.argv[1], rep=int(sys.argv[2]), config=sys.argv[3], log2_n=int(sys.argv[4]), kernel=r["kernel"],
                      avg_launch_us=round(r["avg_launch_us"], 3), frac=r["frac"], hbm_spec_ratio=r.get("hbm_spec_ratio"))))
PY
    done
    FISHING_HIP_LIB="$lib" timeout -k 10 300 python3 scripts/exp/run_f_rows.py > "$O/rows_${var}_$rep.jsonl" 2> "$O/rows.err" || { echo "$var rows failed"; tail -5 "$O/rows.err"; exit 3; }
  done
  echo "rep $rep done"
done
echo done
