#!/bin/bash
# On the GPU box: one line of gpurun_out/box_state.jsonl = which GPU this is (unique id), its clocks / temperatures / power
# state while stepping, and the float64 N = 2^24 step time (the one figure that reads 129 us on some boxes and 137 us on
# others with the same library: DESIGN.md section 5).  Cheap (~10 s): appended to the round's ordinary GPU sessions.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
O="$REPO/gpurun_out"; mkdir -p "$O"
cd "$REPO"
python3 bench.py --f64 --n-envs 16777216 --steps 4000 --warmup 500 --no-cpu-baseline --no-subrecords > /tmp/box_f64.json 2> /tmp/box_f64.err &
BPID=$!
: > /tmp/box_smi.jsonl     # one sample per half second while the bench runs; the summary keeps the one with the highest sclk
while kill -0 $BPID 2>/dev/null; do
  rocm-smi --showuniqueid --showclocks --showtemp --showpower --showperflevel --json 2>/dev/null | tr -d '\n' >> /tmp/box_smi.jsonl
  echo >> /tmp/box_smi.jsonl
  sleep 0.5
done
wait $BPID
python3 bench.py --n-envs 67108864 --steps 100 --warmup 20 --no-cpu-baseline --no-subrecords > /tmp/box_f32.json 2> /tmp/box_f32.err
python3 - <<'P' >> "$O/box_state.jsonl"
import json, time
def line(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
f64, f32 = line("/tmp/box_f64.json"), line("/tmp/box_f32.json")
samples = []
for ln in open("/tmp/box_smi.jsonl"):
    try:
        samples.append(json.loads(ln).get("card0", {}))
    except Exception:
        pass
mhz = lambda s, k: int("".join(c for c in str(s.get(k, "0")) if c.isdigit()) or 0)
smi = max(samples, key=lambda s: mhz(s, "sclk clock speed:")) if samples else {}
smi = {"card0": dict(smi, n_samples=len(samples), fclk_seen=sorted({mhz(s, "fclk clock speed:") for s in samples}),
                     mclk_seen=sorted({mhz(s, "mclk clock speed:") for s in samples}))}
print(json.dumps({"time": int(time.time()), "smi": smi.get("card0", smi),
                  "f64_2p24_avg_launch_us": f64.get("roofline", {}).get("avg_launch_us"),
                  "f32_2p26_avg_launch_us": f32.get("roofline", {}).get("avg_launch_us")}))
P
tail -1 "$O/box_state.jsonl"
