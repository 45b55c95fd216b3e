"""fishing-v11 float32 at N = 2^25 (a zig-zag size): which kernel the dispatch picks and how long a step takes."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, gym_fishing_amd as gf
n = 1 << 25
acts = bench.make_actions(torch, bench.CONFIGS["v1"], n, 2)
env = gf.make("fishing-v11", num_envs=n, seed=1)
for d in env.model_params.values():
    d["sigma"] = 0.1
env.reset(); env.step_many(acts, 16)
us = min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2))
print(json.dumps({"us": round(us, 1), "kernel": env.step_kernel_name(acts[0])}))
