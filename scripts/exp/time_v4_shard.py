"""fishing-v4 derived + sigma array, returns, at its config-5 shard (N = 2^21) and at 2^22 / 2^24: us per step by HIP events."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import gym_fishing_amd as gf
for ln in (21, 22, 24):
    n = 1 << ln
    env = bench.make_env(gf, torch, "v4", n, 0, True)
    env.reset()
    acts = bench.make_actions(torch, bench.CONFIGS["v4"], n, 8)
    env.step_many(acts, 200)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 400 if ln < 24 else 100); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (400 if ln < 24 else 100))
    us = statistics.median(ts)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FISHING_HIP_LIB", "default")), "log2_n": ln, "us_per_step": round(us, 2),
                      "frac_of_8TBps": round(n * 37 / us / 8e6, 3), "kernel": env.step_kernel_name(acts[0])}), flush=True)
    del env, acts
    torch.cuda.empty_cache()
