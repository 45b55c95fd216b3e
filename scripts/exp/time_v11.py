"""fishing-v11 (growth function per env) step at N = 2^22, float32 / float64."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 0.4 - 1.0)
for dtype, auto in ((torch.float32, True), (torch.float32, False), (torch.float64, True)):
    env = gf.make("fishing-v11", num_envs=n, seed=1, dtype=dtype, auto_reset=auto)
    for d in env.model_params.values():
        d["sigma"] = 0.1
    env.reset(); env.step_many(acts, 100)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    w = 4 if dtype == torch.float32 else 8
    per = 3 * w + 4 + 1 + 8 + 4
    us = statistics.median(ts)
    print(json.dumps({"id": "fishing-v11", "dtype": str(dtype)[6:], "auto_reset": auto, "us": round(us, 2), "bytes_per_env_step": per,
                      "TBps": round(n * per / us / 1e6, 2)}), flush=True)
    del env
for policy, param in (("random", 0.0), ("escapement", 0.5)):
    env = gf.make("fishing-v11", num_envs=n, seed=1)
    for d in env.model_params.values():
        d["sigma"] = 0.1
    env.reset(); env.rollout(101, policy=policy, param=param); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(505, policy=policy, param=param); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts)
    print(json.dumps({"id": "fishing-v11", "rollout": policy, "env_steps_per_s": "%.3e" % (n * 505 / ms * 1e3)}), flush=True)
# caller-driven steps: a launch per step vs fishing_step_fused_f32 (101 steps per launch), launch-bound sizes
for ln in (19, 20):
    nn = 1 << ln
    ring2 = torch.empty((8, nn + 3072), device="cuda"); a2 = ring2[:, :nn]; a2.copy_(torch.rand((8, nn), device="cuda") * 0.4 - 1.0)
    env = gf.make("fishing-v11", num_envs=nn, seed=1, track_returns=True)
    for d in env.model_params.values():
        d["sigma"] = 0.1
    env.reset(); env.step_many(a2, 202); env.step_many(a2, 101, fused=True); torch.cuda.synchronize()
    res = {}
    for fused in (False, True):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                env.step_many(a2, 101, fused=fused)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 404)
        res["fused" if fused else "per_step"] = round(statistics.median(ts), 3)
    print(json.dumps({"id": "fishing-v11", "log2_n": ln, "us_per_step": res,
                      "env_steps_per_s_fused": "%.3e" % (nn / res["fused"] * 1e6)}), flush=True)
    del env
