#!/bin/bash
set -u
O=gpurun_out/r02_s6
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
echo "== profile v4_24"; bash scripts/profile_bench.sh r02_s6/prof_v4_24 --config v4 --n-envs 16777216; echo rc=$?
echo "== profile v4_21"; bash scripts/profile_bench.sh r02_s6/prof_v4_21 --config v4; echo rc=$?
timeout -k 10 400 python bench.py --config v4 > $O/bench_v4.json 2> $O/bench_v4.err; echo rc=$?
timeout -k 10 300 python bench.py --config v4 --n-envs 16777216 --no-cpu-baseline --no-subrecords --steps 1010 --warmup 101 > $O/bench_v4_2p24.json 2> $O/bench_v4_2p24.err; echo rc=$?
for v in base fusedkeys0; do echo "== fused $v"; FISHING_HIP_LIB=gym_fishing_amd/_lib/variants/libfishing_hip_$v.so timeout -k 10 300 python scripts/exp/time_fused.py 2>&1 | tail -1 | tee $O/fused_$v.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_s6/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-22s value %.3e avg_us %.2f frac %.3f %s" % (f.split("/")[-1], d["value"], r["avg_launch_us"], r["frac"], r["kernel"]))
    for k in ("bare_step","hbm_resident"):
        if k in d: print("    %s: us %.2f frac %.3f" % (k, d[k]["avg_launch_us"], d[k]["frac"]))
PY
