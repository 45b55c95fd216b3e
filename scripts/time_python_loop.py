#!/usr/bin/env python3
"""Host overhead of the Python env.step() loop (the drop-in use case) vs the C-enqueued loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf

for ln in (16, 20, 22):
    n = 1 << ln
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
    env.reset()
    acts = torch.rand((8, n), device="cuda") * 2 - 1
    for k in range(50):
        env.step(acts[k % 8])
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for k in range(K):
        env.step(acts[k % 8])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    env.step_many(acts, K)
    torch.cuda.synchronize()
    t_c = time.perf_counter() - t0
    print("N=2^%d  python loop: enqueue %.2f us/step, wall %.2f us/step (%.3e env-steps/s) | step_many wall %.2f us/step"
          % (ln, t_enq / K * 1e6, t_all / K * 1e6, n * K / t_all, t_c / K * 1e6), flush=True)
