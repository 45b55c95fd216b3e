"""Per-step launches at the launch-bound sizes (N = 2^18 .. 2^22), fishing-v1 bare / with returns, per build variant."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
res = {}
for ln in (18, 19, 20, 21, 22):
    n = 1 << ln
    ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    for ret in (False, True):
        env = gf.make("fishing-v1", num_envs=n, seed=1, sigma=0.1, track_returns=ret); env.reset()
        env.step_many(acts, 500)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 500); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 2)
        res["2^%d%s" % (ln, "_ret" if ret else "")] = round(statistics.median(ts), 2)
        del env
    del ring, acts
print(json.dumps(res))
