"""fishing-v1 with a K that is not a power of two (true division: the catch-all kernel) next to K = 1, N = 2^22."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
for K in (1.0, 1.5):
    for ret in (False, True):
        env = gf.make("fishing-v1", sigma=0.1, K=K, init_state=0.75 * K, num_envs=n, seed=1, track_returns=ret)
        env.reset(); env.step_many(acts, 100)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 5)
        print(json.dumps({"K": K, "returns": ret, "us": round(statistics.median(ts), 2), "kernel": env.step_kernel_name(acts[0])}), flush=True)
