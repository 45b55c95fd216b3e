#!/bin/bash
# GPU session 5 of round 2 (final build): rocprofv3 kernel stats + PMC passes per workload, SQ passes, bench lines.
set -u
O=gpurun_out/r02_s5
mkdir -p $O
PART="${1:-all}"
if [ "$PART" = "a" ] || [ "$PART" = "all" ]; then
for spec in "v1:" "v1_bare:--no-returns" "v0:--config v0" "v2:--config v2" "v1_2p26:--n-envs 67108864"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== profile $tag ($flags)"; bash scripts/profile_bench.sh r02_s5/prof_$tag $flags; echo rc=$?
done
fi
if [ "$PART" = "b" ] || [ "$PART" = "all" ]; then
for spec in "v4_21:--config v4" "v4_24:--config v4 --n-envs 16777216" "v4s_21:--config v4 --v4-stored" "v4s_24:--config v4 --v4-stored --n-envs 16777216" "v4_21_bare:--config v4 --no-returns"; do
  tag=${spec%%:*}; flags=${spec#*:}
  echo "== profile $tag ($flags)"; bash scripts/profile_bench.sh r02_s5/prof_$tag $flags; echo rc=$?
done
echo "== SQ v4"; bash scripts/profile_sq.sh r02_s5/sq_v4 --config v4 --no-returns; echo rc=$?
echo "== SQ v1"; bash scripts/profile_sq.sh r02_s5/sq_v1 --no-returns; echo rc=$?
fi
if [ "$PART" = "c" ] || [ "$PART" = "all" ]; then
echo "== driver command"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc=$?
echo "== default"; timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
for c in v0 v2 v4; do echo "== config $c"; timeout -k 10 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo rc=$?; done
timeout -k 10 300 python bench.py --config v4 --n-envs 16777216 --no-cpu-baseline --no-subrecords --steps 1010 --warmup 101 > $O/bench_v4_2p24.json 2> $O/bench_v4_2p24.err; echo rc=$?
timeout -k 10 300 python bench.py --config v4 --v4-stored --no-cpu-baseline --no-subrecords > $O/bench_v4_stored.json 2> $O/bench_v4_stored.err; echo rc=$?
timeout -k 10 300 python bench.py --no-returns --no-cpu-baseline --no-subrecords > $O/bench_v1_bare.json 2> $O/bench_v1_bare.err; echo rc=$?
timeout -k 10 300 python bench.py --compact --no-cpu-baseline > $O/bench_compact.json 2> $O/bench_compact.err; echo rc=$?
timeout -k 10 300 python bench.py --compact --no-returns --no-cpu-baseline > $O/bench_compact_bare.json 2> $O/bench_compact_bare.err; echo rc=$?
timeout -k 10 400 python bench.py --no-cpu-baseline --no-subrecords --extra > $O/bench_extra.json 2> $O/bench_extra.err; echo rc=$?
FISHING_BENCH_BACKEND=gloo FISHING_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 --n-envs 1048576 --no-cpu-baseline > $O/bench_2rank_gloo_rehearsal.json 2> $O/bench_2rank.err; echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_s5/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r=d["roofline"]
    print("%-34s gpus %d value %.3e ms/step %.4f avg_us %.2f region_us %.2f frac %.3f B %s %s" % (f.split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], r["avg_launch_us"], r["avg_launch_us_timed_region"], r["frac"], r["bytes_per_env_step"], r["kernel"]))
    for k in ("bare_step","hbm_resident"):
        if k in d: print("    %s: us %.2f frac %.3f" % (k, d[k]["avg_launch_us"], d[k]["frac"]))
    if "fused_step_many" in d:
        for kk,v in d["fused_step_many"].items():
            if isinstance(v,dict): print("    fused %s: per-step %.2f us (%.2e) | fused+rows %.2f us (%.2e) | fused %.2f us (%.2e)" % (kk, v["per_step_launches"]["us_per_step"], v["per_step_launches"]["env_steps_per_s"], v["fused_with_reward_done_rows"]["us_per_step"], v["fused_with_reward_done_rows"]["env_steps_per_s"], v["fused_last_step_outputs_only"]["us_per_step"], v["fused_last_step_outputs_only"]["env_steps_per_s"]))
    if "extra" in d: print("    extra:", json.dumps(d["extra"]))
    cb=d.get("cpu_baseline")
    if cb: print("    cpu: scalar %.3e | all-cores %s | numpy-vec %s | C all %s" % (cb["value"], cb.get("python_port_all_cores",{}).get("value"), cb.get("numpy_vectorised",{}).get("value"), cb.get("c_port_all_cores",{}).get("value")))
PY
fi
