"""A/B build variants on the general-kernel workloads (fp64 v1, fp32 zoo v9, fp32 v1 with done_bits)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
n = 1 << 22
acts = torch.rand((4, n), device="cuda") * 0.5 - 1
res = {}
for key, env_id, kw in (("v1_f64", "fishing-v1", dict(dtype=torch.float64)), ("v9_f32", "fishing-v9", {}), ("v1_f32_bits", "fishing-v1", dict(done_bits=True))):
    env = gf.make(env_id, sigma=0.1, num_envs=n, seed=1, **kw); env.reset(); env.step_many(acts, 100)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    res[key] = round(statistics.median(ts), 2)
    del env
print(json.dumps(res))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
        p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FISHING_HIP_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(rnd, tag, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "ERR " + p.stderr[-200:], flush=True)
