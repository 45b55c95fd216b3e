"""What the ragged tail (N not a multiple of 1024: a second, one-workgroup launch of the general kernel per step) costs."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for n in (1 << 16, (1 << 16) + 8, 1_000_000, 1 << 20, (1 << 20) + 8, 1 << 22, (1 << 22) + 8):
    stride = (n + 3075) // 4 * 4
    ring = torch.empty((8, stride), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
    env.reset(); env.step_many(acts, 200)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 400); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 2.5)
    print(json.dumps({"n": n, "tail": n % 1024, "us_per_step": round(statistics.median(ts), 2)}), flush=True)
    del env, ring, acts
