"""In-kernel-policy rollouts at N = 2^22 (505 steps per launch): env-steps/s by HIP events."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
for idn, pol, param in (("fishing-v1", "random", 0.0), ("fishing-v1", "escapement", 0.5), ("fishing-v0", "random", 0.0), ("fishing-v2", "random", 0.0),
                        ("fishing-v4", "random", 0.0), ("fishing-v4", "escapement", 0.5)):
    env = gf.make(idn, num_envs=n, seed=1, sigma=0.1, track_returns=True)
    env.reset()
    env.rollout(101, policy=pol, param=param)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(505, policy=pol, param=param); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")), "id": idn, "policy": pol, "ms": round(ms, 3),
                      "env_steps_per_s": "%.4g" % (n * 505 / ms * 1e3)}), flush=True)
    del env
# the fused K-step kernel on fishing-v4 (caller's actions: a quarter of the envs finishes per step)
for ln in (20, 22):
    nn = 1 << ln
    env = gf.make("fishing-v4", num_envs=nn, seed=1, sigma=0.05, track_returns=True)
    env.reset()
    a2 = torch.empty((8, nn + 3072), device="cuda")[:, :nn]
    a2.copy_(torch.rand((8, nn), device="cuda") * 2 - 1)
    env.step_many(a2, 101, fused=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(a2, 101, fused=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")), "id": "fishing-v4", "policy": "fused step_many, 2^%d" % ln,
                      "ms": round(ms, 3), "env_steps_per_s": "%.4g" % (nn * 101 / ms * 1e3)}), flush=True)
    del env, a2
