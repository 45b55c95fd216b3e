#!/usr/bin/env python3
"""A random-policy rollout the way an RL loop drives it: actions come from the caller every
step (here torch.rand on the device), observations / rewards / dones come back as device
tensors, finished envs are reset inside the kernel (SB3 VecEnv semantics)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf  # noqa: E402

n = 1 << 22
env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=0, track_returns=True)
obs = env.reset()
steps = 303
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    actions = torch.rand(n, device="cuda") * 2 - 1          # your policy goes here
    obs, reward, done, info = env.step(actions)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%.3e env-steps/s incl. the torch.rand policy" % (n * steps / dt))
print(env.episode_stats())
