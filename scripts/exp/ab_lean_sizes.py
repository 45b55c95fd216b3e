"""A/B of lean-kernel build variants (gym_fishing_amd/_lib/variants/*.so) across batch sizes: bare fishing-v1 step,
padded action ring of 4, one process per variant and size."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
res = {}
for ln in (22, 24, 26):
    n = 1 << ln
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1); env.reset()
    K = 400 if ln == 22 else (100 if ln == 24 else 30)
    env.step_many(acts, K)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, K); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
    res["2^%%d" %% ln] = round(statistics.median(ts), 2)
    del env, ring, acts
    torch.cuda.empty_cache()
print(json.dumps(res))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
        p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FISHING_HIP_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(rnd, tag, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "ERR " + p.stderr[-300:], flush=True)
