"""The fused K-step kernel on fishing-v0 (caller's indices), N = 2^20 / 2^22, 101 steps per launch: env-steps/s by HIP events."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for ln in (20, 22):
    nn = 1 << ln
    env = gf.make("fishing-v0", num_envs=nn, seed=1, sigma=0.1, track_returns=True)
    env.reset()
    a2 = torch.randint(0, 100, (8, nn), device="cuda", dtype=torch.int32)
    env.step_many(a2, 101, fused=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(a2, 101, fused=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")), "id": "fishing-v0", "policy": "fused step_many, 2^%d" % ln,
                      "ms": round(ms, 3), "env_steps_per_s": "%.4g" % (nn * 101 / ms * 1e3)}), flush=True)
    del env, a2
