#!/bin/bash
# GPU session 3b of round 2: rocprofv3 profiles of the fishing-v4 workloads (derived and stored parameters, 2^21 and
# 2^24), one SQ-counter pass, the driver's bench command and the full default line.
set -u
O=gpurun_out/r02_s3
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_r02.py -m gpu -q -p no:cacheprovider > $O/tests_b.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_b.log
echo "== A/B record variants"; timeout -k 10 600 python scripts/exp/ab_r02.py base recsplit nolatch > $O/ab_rec.jsonl 2> $O/ab_rec.err; cat $O/ab_rec.jsonl
echo "== A/B large N"; timeout -k 10 600 python scripts/exp/ab_r02_large.py base blocked > $O/ab_large.jsonl 2> $O/ab_large.err; cat $O/ab_large.jsonl
for spec in "v4_21:--config v4" "v4_24:--config v4 --n-envs 16777216" "v4s_21:--config v4 --v4-stored" "v4s_24:--config v4 --v4-stored --n-envs 16777216"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== profile $tag ($flags)"; bash scripts/profile_bench.sh r02_s3/prof_$tag $flags; echo rc=$?
done
echo "== SQ v4"; bash scripts/profile_sq.sh r02_s3/sq_v4 --config v4 --no-returns; echo rc=$?
echo "== SQ v1"; bash scripts/profile_sq.sh r02_s3/sq_v1 --no-returns; echo rc=$?
echo "== driver command"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$?
echo "== default"; timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_s3/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-22s value %.3e ms/step %.4f avg_us %.2f region_us %.2f frac %.3f %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"], r["avg_launch_us_timed_region"], r["frac"], r["kernel"]))
PY
