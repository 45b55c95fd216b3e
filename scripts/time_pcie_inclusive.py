#!/usr/bin/env python3
"""PCIe-inclusive rate: the caller hands HOST action buffers every step (never the bench `value`)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf
n = 1 << 22
env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1)
env.reset()
host_np = np.random.default_rng(0).uniform(-1, 1, (4, n)).astype(np.float32)
pinned = torch.from_numpy(host_np).pin_memory()
for name, src in (("pageable numpy", [host_np[k] for k in range(4)]), ("pinned torch", [pinned[k] for k in range(4)])):
    for k in range(8):
        env.step(src[k % 4])
    torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for k in range(K):
        env.step(src[k % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s actions: %.1f us/step, %.3e env-steps/s (H2D %.1f GB/s incl. the step)" % (name, dt / K * 1e6, n * K / dt, n * 4 * K / dt / 1e9))
