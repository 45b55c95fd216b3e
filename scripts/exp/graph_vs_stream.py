"""Launch-bound regime (N <= 2^20): K dependent step launches enqueued by one C call on a stream vs
the same K launches captured in a hipGraph (device-resident step counter + a counter-bump kernel per
step)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
from gym_fishing_amd.graphs import GraphedSteps
K = 200
for log2n in (14, 16, 18, 20, 22):
    n = 1 << log2n
    ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    res = {"log2n": log2n}
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1); env.reset(); env.step_many(acts, K)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); env.step_many(acts, K * 5); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / (K * 5) * 1e6)
    res["stream_us_per_step"] = round(min(ts), 2)
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1); env.reset()
    g = GraphedSteps(env, acts, n_steps=K)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / (K * 5) * 1e6)
    res["graph_us_per_step"] = round(min(ts), 2)
    print(json.dumps(res), flush=True)
