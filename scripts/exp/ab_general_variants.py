"""A/B build variants on the general-kernel workloads (fp64 v4, fp32 v11, fp32 v1 with done_bits, fp32 v1 with
the terminal-observation record)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
n = 1 << 22
acts = torch.rand((4, n), device="cuda") * 0.5 - 1
res = {}
for key, env_id, kw in (("v4_f64", "fishing-v4", dict(dtype=torch.float64)), ("v11_f32", "fishing-v11", {}), ("v1_f32_bits", "fishing-v1", dict(done_bits=True)),
                        ("v1_f32_term", "fishing-v1", dict(record_terminal_obs=True))):
    if env_id != "fishing-v11":
        kw = dict(kw, sigma=0.1)
    env = gf.make(env_id, num_envs=n, seed=1, **kw); env.reset(); env.step_many(acts, 100)
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
