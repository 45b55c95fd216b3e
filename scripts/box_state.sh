#!/bin/bash
# On the GPU box: one line of gpurun_out/box_state.jsonl = which GPU this is (unique id), its clocks / temperatures / power
# state while stepping, and the float64 N = 2^24 step time (the one figure that reads 129 us on some boxes and 137 us on
# others with the same library: DESIGN.md section 5).  Cheap (~10 s): appended to the round's ordinary GPU sessions.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out"; mkdir -p "$O"
cd "$REPO"
python3 bench.py --f64 --n-envs 16777216 --steps 200 --warmup 50 --no-cpu-baseline --no-subrecords > /tmp/box_f64.json 2> /tmp/box_f64.err &
BPID=$!
sleep 4     # (sample while the bench is stepping)
rocm-smi --showuniqueid --showclocks --showtemp --showpower --showperflevel --json > /tmp/box_smi.json 2>/dev/null || echo '{}' > /tmp/box_smi.json
wait $BPID
python3 bench.py --n-envs 67108864 --steps 100 --warmup 20 --no-cpu-baseline --no-subrecords > /tmp/box_f32.json 2> /tmp/box_f32.err
python3 - <<'P' >> "$O/box_state.jsonl"
import json, time
def line(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
f64, f32, smi = line("/tmp/box_f64.json"), line("/tmp/box_f32.json"), line("/tmp/box_smi.json")
print(json.dumps({"time": int(time.time()), "smi": smi.get("card0", smi),
                  "f64_2p24_avg_launch_us": f64.get("roofline", {}).get("avg_launch_us"),
                  "f32_2p26_avg_launch_us": f32.get("roofline", {}).get("avg_launch_us")}))
P
tail -1 "$O/box_state.jsonl"
