#!/bin/bash
# GPU session 1 of round 2: new tests, the driver's bench command, per-config lines, A/B against the round-1 tree
# (a `git worktree add --detach .r01_baseline <round-1 head>` with its own built library, removed at the end of the round).
set -u
O=gpurun_out/r02_s1
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_r02.py tests/test_gpu_bench.py -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
echo "== driver command"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$?
echo "== default"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
echo "== r01 baseline tree, same box"; timeout -k 10 300 python .r01_baseline/bench.py --no-cpu-baseline > $O/bench_r01_default.json 2> $O/bench_r01.err; echo rc=$?
timeout -k 10 300 python .r01_baseline/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_r01_driver.json 2>> $O/bench_r01.err; echo rc=$?
for c in v0 v2 v4; do echo "== config $c"; timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo rc=$?; done
echo "== v4 stored"; timeout -k 10 300 python bench.py --config v4 --v4-stored --no-cpu-baseline > $O/bench_v4_stored.json 2> $O/bench_v4_stored.err; echo rc=$?
echo "== v4 2^24"; timeout -k 10 300 python bench.py --config v4 --n-envs 16777216 --no-cpu-baseline --no-subrecords --steps 1010 --warmup 101 > $O/bench_v4_2p24.json 2> $O/bench_v4_2p24.err; echo rc=$?
timeout -k 10 300 python bench.py --config v4 --v4-stored --n-envs 16777216 --no-cpu-baseline --no-subrecords --steps 1010 --warmup 101 > $O/bench_v4_2p24_stored.json 2>> $O/bench_v4_2p24.err; echo rc=$?
echo "== extra"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords --extra > $O/bench_extra.json 2> $O/bench_extra.err; echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_s1/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]
    print("%-28s value %.3e  ms/step %.4f  avg_us %.2f  med_us %s  frac %.3f  B %s  %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"], r.get("avg_launch_us_timed_region"), r["frac"], r.get("bytes_per_env_step"), r.get("kernel")))
    for k in ("bare_step","hbm_resident"):
        if k in d: print("    %s: us %.2f frac %.3f" % (k, d[k]["avg_launch_us"], d[k]["frac"]))
    if "fused_step_many" in d:
        for kk,v in d["fused_step_many"].items():
            if isinstance(v,dict): print("    fused %s: per-step %.2f us (%.2e) | fused+rows %.2f us (%.2e) | fused %.2f us (%.2e)" % (kk, v["per_step_launches"]["us_per_step"], v["per_step_launches"]["env_steps_per_s"], v["fused_with_reward_done_rows"]["us_per_step"], v["fused_with_reward_done_rows"]["env_steps_per_s"], v["fused_last_step_outputs_only"]["us_per_step"], v["fused_last_step_outputs_only"]["env_steps_per_s"]))
    if "extra" in d: print("    extra:", json.dumps(d["extra"]))
PY
