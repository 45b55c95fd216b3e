"""A/B the lean-kernel scheduling knobs (variant libraries) on the bench workloads, one process each,
two interleaved rounds."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
res = {}
for ret in (False, True):
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=ret); env.reset(); env.step_many(acts, 300)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 400); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 2.5)
    res["ret" if ret else "bare"] = round(statistics.median(ts), 2)
print(json.dumps(res))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
        p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FISHING_HIP_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(rnd, tag, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "ERR " + p.stderr[-200:], flush=True)
