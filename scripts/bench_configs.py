#!/usr/bin/env python3
"""Kernel-only timing of every BASELINE.json config on one GPU (per-GPU shard sizes for the
8-GPU configs).  Supplementary to bench.py (which measures the metric's config)."""
import json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf


def timed(env, acts, steps):
    env.step_many(acts, min(steps, 64))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / steps)
    return statistics.median(ts)


rows = []
# config 1: single env, scalar protocol (plumbing): Python-visible steps/s
env = gf.make("fishing-v1", sigma=0.0)
env.reset()
a = np.array([-0.9375], dtype=np.float32)
t0 = time.perf_counter()
k = 0
for _ in range(3):
    done = False
    env.reset()
    while not done:
        _, _, done, _ = env.step(a)
        k += 1
rows.append({"config": "1: fishing-v1 sigma=0, single env, scalar protocol", "env_steps_per_s": round(k / (time.perf_counter() - t0))})
for tag, env_id, n, kw, per, act in (
        ("2: fishing-v1 sigma=0.1, N=2^20", "fishing-v1", 1 << 20, dict(sigma=0.1), 25, "cts"),
        ("3: fishing-v0 n_actions=100, N=2^22", "fishing-v0", 1 << 22, dict(sigma=0.1, n_actions=100), 25, "int"),
        ("4: fishing-v2 C=0.5, N=2^22 on one GPU", "fishing-v2", 1 << 22, dict(sigma=0.1, C=0.5), 25, "low"),
        ("4: fishing-v2, per-GPU shard 2^19 of the 8-GPU run", "fishing-v2", 1 << 19, dict(sigma=0.1, C=0.5), 25, "low"),
        ("5: fishing-v4 (r,K,sigma arrays), per-GPU shard 2^21 of N=2^24", "fishing-v4", 1 << 21, dict(sigma="arr", sigma_p=0.1), 37, "cts"),
        ("5: fishing-v4 (r,K,sigma arrays), N=2^24 on one GPU", "fishing-v4", 1 << 24, dict(sigma="arr", sigma_p=0.1), 37, "cts")):
    if kw.get("sigma") == "arr":
        kw = dict(kw, sigma=torch.full((n,), 0.05))
    env = gf.make(env_id, num_envs=n, seed=1, **kw)
    env.reset()
    if act == "int":
        acts = torch.randint(0, 100, (4, n), device="cuda", dtype=torch.int32)
    elif act == "low":
        acts = torch.rand((4, n), device="cuda") * 0.2 - 1.0
    else:
        acts = torch.rand((4, n), device="cuda") * 2 - 1
    us = timed(env, acts, 200 if n <= (1 << 22) else 50)
    rows.append({"config": tag, "us_per_launch": round(us, 2), "bytes_per_env_step": per, "GBps": round(n * per / us / 1e3),
                 "env_steps_per_s": "%.3e" % (n / us * 1e6)})
    del env, acts
    torch.cuda.empty_cache()
for r in rows:
    print(json.dumps(r), flush=True)
