"""fishing_step_fused_* vs per-step launches at the launch-bound sizes, per build variant (FISHING_HIP_LIB)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
res = {}
for env_id, kw in (("fishing-v1", dict(sigma=0.1)), ("fishing-v2", dict(sigma=0.1)), ("fishing-v4", dict(sigma=0.05))):
    for ln in (18, 19, 20, 22):
        n = 1 << ln
        env = gf.make(env_id, num_envs=n, seed=1, track_returns=True, **kw); env.reset()
        acts = torch.rand((8, n), device="cuda") * 2 - 1
        rows_r = torch.empty((101, n), device="cuda"); rows_d = torch.empty((101, n), dtype=torch.uint8, device="cuda")
        out = {}
        for tag, call in (("launches", lambda: env.step_many(acts, 101)),
                          ("fused_rows", lambda: env.step_many(acts, 101, fused=True, rewards_out=rows_r, dones_out=rows_d)),
                          ("fused", lambda: env.step_many(acts, 101, fused=True))):
            call()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 101)
            out[tag] = round(statistics.median(ts), 2)
        res["%s 2^%d" % (env_id[-2:], ln)] = out
        del env, acts, rows_r, rows_d; torch.cuda.empty_cache()
print(json.dumps(res))
