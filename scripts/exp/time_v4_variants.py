import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
acts = torch.rand((4, n), device="cuda") * 0.3 - 1.0
for tag, sig in (("scalar sigma (lean)", 0.05), ("sigma array (lean SIGARR)", torch.full((n,), 0.05))):
    env = gf.make("fishing-v4", sigma=sig, num_envs=n, seed=1)
    env.reset(); env.step_many(acts, 40)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 100); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10)
    print(tag, round(statistics.median(ts), 2), "us")
