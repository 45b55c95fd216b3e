#!/usr/bin/env python3
"""Does the size of the resident action ring (cache residency of the action stream) matter?"""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf
n = 1 << 22
for ret in (False, True):
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=ret)
    env.reset()
    res = {}
    rings = {}
    for R in (1, 2, 4, 8, 16, 32):
        buf = torch.empty((R, n + 3072), device="cuda")
        a = buf[:, :n]
        a.copy_(torch.rand((R, n), device="cuda") * 2 - 1)
        rings[R] = a
        res[R] = []
    for rnd in range(5):
        for R, a in rings.items():
            env.step_many(a, 64)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(a, 320); e1.record(); torch.cuda.synchronize()
            res[R].append(e0.elapsed_time(e1) * 1e3 / 320)
    for R in rings:
        us = statistics.median(res[R])
        per = 33 if ret else 25
        print(json.dumps({"returns": ret, "ring": R, "ring_MB": round(R * n * 4 / 1e6), "med_us": round(us, 2), "GBps": round(n * per / us / 1e3)}), flush=True)
    del env, rings
    torch.cuda.empty_cache()
