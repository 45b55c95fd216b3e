import json, statistics, sys, torch
root = sys.argv[1]
sys.path.insert(0, root)
import gym_fishing_amd as gf
res = {}
for key, dtype, ret in (("f64", torch.float64, False), ("f64_ret", torch.float64, True), ("f32", torch.float32, False)):
    n = 1 << 22
    env = gf.make("fishing-v1", num_envs=n, seed=1, sigma=0.1, dtype=dtype, track_returns=ret); env.reset()
    ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    env.step_many(acts, 200)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    res[key] = round(statistics.median(ts), 2)
    del env, ring, acts; torch.cuda.empty_cache()
print(root, json.dumps(res))
