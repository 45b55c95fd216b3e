#!/bin/bash
# GPU session 2 of round 2: full GPU suite on the current build, build-variant A/B, driver-command bench.
set -u
O=gpurun_out/r02_s2
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
echo "== A/B"; timeout -k 10 900 python scripts/exp/ab_r02.py > $O/ab.jsonl 2> $O/ab.err; echo rc=$?; cat $O/ab.jsonl
echo "== driver command"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-subrecords > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_s2/bench_driver.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("driver: value %.3e ms/step %.4f avg_us %.2f region_us %.2f frac %.3f" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["avg_launch_us_timed_region"], r["frac"]))
PY
