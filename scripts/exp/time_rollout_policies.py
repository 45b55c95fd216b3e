"""Fused in-kernel-policy rollout rate (env-steps/s) per build variant (FISHING_HIP_LIB): fishing-v1 / v2 / v4, N = 2^22."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
res = {}
n = 1 << 22
for env_id, kw in (("fishing-v1", dict(sigma=0.1)), ("fishing-v2", dict(sigma=0.1)), ("fishing-v4", dict(sigma=0.05)), ("fishing-v9", dict(sigma=0.1))):
    for pol, param in (("random", 0.0), ("escapement", 0.5)):
        env = gf.make(env_id, num_envs=n, seed=1, track_returns=True, **kw); env.reset()
        env.rollout(101, policy=pol, param=param); torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); env.rollout(1010, policy=pol, param=param); torch.cuda.synchronize()
            best = max(best, n * 1010 / (time.perf_counter() - t0))
        res["%s %s" % (env_id[-2:], pol)] = "%.3e" % best
        del env
print(json.dumps(res))
