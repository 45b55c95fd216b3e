"""Round-2 A/B at HBM-resident sizes (N = 2^24, 2^26): build variants x workgroup caps, kernel-only."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, statistics, sys, torch
sys.path.insert(0, %r)
import gym_fishing_amd as gf
res = {}
for ln, ret, steps in ((24, False, 100), (26, False, 40), (26, True, 40)):
    n = 1 << ln
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    for cap in (0, 2048, 1024, 512):
        env = gf.make("fishing-v1", num_envs=n, seed=1, sigma=0.1, track_returns=ret, launch_blocks=cap); env.reset()
        env.step_many(acts, steps)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / steps)
        res["2^%%d%%s cap%%d" %% (ln, "_ret" if ret else "", cap)] = round(statistics.median(ts), 1)
        del env
    del ring, acts; torch.cuda.empty_cache()
print(json.dumps(res))
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, "gym_fishing_amd", "_lib", "variants", "*.so")))
only = sys.argv[1:]
for lib in libs:
    tag = os.path.basename(lib)[len("libfishing_hip_"):-3]
    if only and tag not in only:
        continue
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FISHING_HIP_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(json.dumps({"variant": tag, "us": json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else "ERR " + p.stderr[-300:]}), flush=True)
