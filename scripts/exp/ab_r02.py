"""Round-2 A/B of build variants (libraries under gym_fishing_amd/_lib/variants/, one process per variant and round,
interleaved): the bench workloads of every config, kernel-only (HIP events around step_many), padded action ring."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
res = {}
def run(key, env_id, ln, ret=False, dtype=torch.float32, lo=-1.0, hi=1.0, steps=400, **kw):
    n = 1 << ln
    if kw.get("sigma") == "arr":
        kw["sigma"] = torch.full((n,), 0.05, device="cuda")
    env = gf.make(env_id, num_envs=n, seed=1, track_returns=ret, dtype=dtype, **kw); env.reset()
    ring = torch.empty((8, n + 3072), device="cuda", dtype=torch.int32 if env_id == "fishing-v0" else torch.float32)
    acts = ring[:, :n]
    if env_id == "fishing-v0": acts.copy_(torch.randint(0, 100, (8, n), device="cuda", dtype=torch.int32))
    else: acts.copy_(torch.rand((8, n), device="cuda") * (hi - lo) + lo)
    env.step_many(acts, steps)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / steps)
    res[key] = round(statistics.median(ts), 2)
    del env, acts, ring; torch.cuda.empty_cache()
run("v1", "fishing-v1", 22, sigma=0.1)
run("v1_ret", "fishing-v1", 22, ret=True, sigma=0.1)
run("v1_21", "fishing-v1", 21, sigma=0.1)
run("v0", "fishing-v0", 22, sigma=0.1)
run("v2", "fishing-v2", 22, sigma=0.1, lo=-1.0, hi=-0.8)
run("v4d_21", "fishing-v4", 21, sigma="arr")
run("v4d_21_ret", "fishing-v4", 21, ret=True, sigma="arr")
run("v4d_24", "fishing-v4", 24, sigma="arr", steps=100)
run("v4s_21", "fishing-v4", 21, sigma="arr", derived_params=False)
run("v9", "fishing-v9", 22, sigma=0.1)
run("v1_term", "fishing-v1", 22, sigma=0.1, record_terminal_obs=True)
run("v1_f64", "fishing-v1", 22, sigma=0.1, dtype=torch.float64, steps=200)
run("v1_24", "fishing-v1", 24, sigma=0.1, steps=100)
print(json.dumps(res))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
only = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
        if only and tag not in only:
            continue
        p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FISHING_HIP_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(json.dumps({"round": rnd, "variant": tag, "us": json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else "ERR " + p.stderr[-300:]}), flush=True)
