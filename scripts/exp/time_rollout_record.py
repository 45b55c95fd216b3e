"""Fused rollout with the [T][4][N] trajectory record (what env.simulate drives): write-bound."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n, T = 1 << 20, 101
for policy, param in (("escapement", 0.5), ("random", 0.0)):
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
    env.reset(); traj = env.rollout(T, policy=policy, param=param, record=True); del traj
    ts = []
    for _ in range(5):
        env.reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); traj = env.rollout(T, policy=policy, param=param, record=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); del traj
    ms = statistics.median(ts)
    print(json.dumps({"policy": policy, "ms": round(ms, 3), "env_steps_per_s": "%.3e" % (n * T / ms * 1e3),
                      "record_write_TBps": round(n * T * 16 / ms / 1e9, 2)}), flush=True)
