#!/usr/bin/env python3
"""Workgroup-size sweep (needs the t1024 variant library: FISHING_HIP_LIB=.../libfishing_hip_t1024.so)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf
n = 1 << 22
acts = torch.rand((8, n), device="cuda") * 2 - 1
shapes = [(4096, 256), (2048, 512), (1024, 1024), (1024, 512), (512, 1024), (4096, 512), (4096, 1024), (768, 1024), (1536, 512)]
for ret in (False, True):
    envs = {}
    for (b, t) in shapes:
        e = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=ret, launch_blocks=b, launch_threads=t)
        e.reset(); e.step_many(acts, 50); envs[(b, t)] = e
    res = {k: [] for k in shapes}
    for rnd in range(5):
        for k, e in envs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 5.0)
    for k in shapes:
        print(json.dumps({"returns": ret, "blocks": k[0], "threads": k[1], "med_us": round(statistics.median(res[k]), 2)}), flush=True)
    del envs
