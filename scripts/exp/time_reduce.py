"""Latency of fishing_reduce_returns (one workgroup over the 4096 per-workgroup record slots)."""
import torch, sys, statistics
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
env = gf.make("fishing-v1", sigma=0.1, num_envs=1 << 20, seed=1, track_returns=True); env.reset()
acts = torch.rand((4, 1 << 20), device="cuda") * 2 - 1
env.step_many(acts, 50); env.episode_stats()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env._lib.fishing_reduce_returns(env._partials.data_ptr(), env._record.data_ptr(), env._stream()); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("reduce_returns us", round(statistics.median(ts), 2), env.episode_stats())
