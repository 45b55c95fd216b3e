#!/bin/bash
# round 5, session 2: full GPU suite after the envs.py / tests split; shape record on the fishing-v4 kernel itself (N = 2^21 and
# 2^24: 128- / 512-thread workgroups, two envs per thread); float64 zoo error record (algebraic form vs the round-trip build)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out/r05_s02"; mkdir -p "$O"
cd "$REPO"
V="$REPO/gym_fishing_amd/_lib/variants"
timeout -k 10 600 python3 -m pytest tests -m gpu -q > "$O/gpu_tests.log" 2>&1 || { tail -30 "$O/gpu_tests.log"; exit 1; }
tail -1 "$O/gpu_tests.log"
: > "$O/v4_shapes.jsonl"
for rep in 1 2 3; do
  for var in product x_tile512 x_tile2048 x_v4e2; do
    for n in 2097152 16777216; do
      lib="$V/libfishing_hip_$var.so"; [ "$var" = product ] && lib="$REPO/gym_fishing_amd/_lib/libfishing_hip.so"
      line=$(FISHING_HIP_LIB="$lib" timeout -k 10 200 python3 bench.py --config v4 --n-envs $n --steps 1010 --warmup 101 --no-subrecords --no-cpu-baseline 2> "$O/shape.err") || { echo "shape $var $n failed"; tail -5 "$O/shape.err"; exit 2; }
      python3 - "$var" "$rep" "$line" >> "$O/v4_shapes.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[3]); r = d["roofline"]
shape = {"product": "1024-env tile, 256 threads x 4 envs", "x_tile512": "512-env tile, 128 threads x 4 envs",
         "x_tile2048": "2048-env tile, 512 threads x 4 envs", "x_v4e2": "1024-env tile, 512 threads x 2 envs"}[sys.argv[1]]
print(json.dumps(dict(variant=sys.argv[1], shape=shape, rep=int(sys.argv[2]), n_envs=d["config"]["envs_per_gpu"], kernel=r["kernel"],
                      avg_launch_us=round(r["avg_launch_us"], 3), frac=r["frac"], hbm_spec_ratio=r.get("hbm_spec_ratio"),
                      episodes=d["episode_stats"]["n_episodes"], mean_return=d["episode_stats"]["mean_return"])))
PY
    done
  done
  echo "shapes rep $rep done"
done
timeout -k 10 600 python3 tests/measure_zoo_f64_error.py --tag algebraic > "$O/zoo_f64_error.jsonl" 2> "$O/zoo_err.err" || { tail -5 "$O/zoo_err.err"; exit 3; }
FISHING_HIP_LIB="$V/libfishing_hip_f64_roundtrip.so" timeout -k 10 600 python3 tests/measure_zoo_f64_error.py --tag round_trip >> "$O/zoo_f64_error.jsonl" 2>> "$O/zoo_err.err" || { tail -5 "$O/zoo_err.err"; exit 4; }
echo done
