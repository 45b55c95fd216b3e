#!/usr/bin/env python3
"""Kernel-only timing of the zoo step kernels (fishing-v5..v11) at N = 2^22, both layouts."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf

n = 1 << 22
acts = torch.rand((4, n), device="cuda") * 0.4 - 1.0
for env_id in ("fishing-v1", "fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10", "fishing-v11"):
    for dtype in (torch.float32, torch.float64):
        kw = {} if env_id == "fishing-v11" else {"sigma": 0.1}
        env = gf.make(env_id, num_envs=n, seed=1, dtype=dtype, **kw)
        if env_id == "fishing-v11":
            for d in env.model_params.values():
                d["sigma"] = 0.1
        env.reset()
        env.step_many(acts, 40)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 100); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10.0)
        w = 4 if dtype == torch.float32 else 8
        per = 2 * w + 4 + w + 1 + 8 + (2 * w if env_id == "fishing-v10" else 0) + (4 if env_id == "fishing-v11" else 0)
        us = statistics.median(ts)
        print(json.dumps({"id": env_id, "dtype": str(dtype)[6:], "us": round(us, 2), "bytes_per_env_step": per,
                          "GBps": round(n * per / us / 1e3), "env_steps_per_s": "%.3e" % (n / us * 1e6)}), flush=True)
        del env
        torch.cuda.empty_cache()
