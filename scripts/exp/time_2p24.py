import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
res = {}
n = 1 << 24
for key, env_id, kw, ret in (("v1", "fishing-v1", dict(sigma=0.1), False), ("v1_ret", "fishing-v1", dict(sigma=0.1), True), ("v4", "fishing-v4", dict(sigma="arr"), False), ("v4_ret", "fishing-v4", dict(sigma="arr"), True)):
    if kw.get("sigma") == "arr": kw = dict(sigma=torch.full((n,), 0.05, device="cuda"))
    env = gf.make(env_id, num_envs=n, seed=1, track_returns=ret, **kw); env.reset()
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    env.step_many(acts, 100)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 100); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10)
    res[key] = round(statistics.median(ts), 1)
    del env, ring, acts; torch.cuda.empty_cache()
print(json.dumps(res))
