import json, os, statistics, sys, torch
sys.path.insert(0, "/root/repo")
import gym_fishing_amd as gf
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 0.4 - 1.0)
for auto, ret in ((True, False), (False, False), (True, True)):
    env = gf.make("fishing-v11", num_envs=n, seed=1, auto_reset=auto, track_returns=ret)
    for d in env.model_params.values():
        d["sigma"] = 0.1
    env.reset(); env.step_many(acts, 100)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    print(json.dumps({"auto_reset": auto, "returns": ret, "us": round(statistics.median(ts), 2), "kernel": env.step_kernel_name(acts[0])}), flush=True)
