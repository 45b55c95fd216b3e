#!/usr/bin/env python3
"""Sweep the workgroup cap of the step kernel at N = 2^22 (interleaved rounds, one process)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
ret = len(sys.argv) > 2 and sys.argv[2] == "ret"
shapes = [(b, t) for t in (256, 128) for b in (1024, 1280, 1365, 1536, 1792, 2048, 2304, 2560, 3072, 4096)]
envs = {}
acts = torch.rand((8, n), device="cuda") * 2 - 1
for (b, t) in shapes:
    e = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=ret, launch_blocks=b, launch_threads=t)
    e.reset()
    e.step_many(acts, 50)
    envs[(b, t)] = e
res = {k: [] for k in shapes}
for rnd in range(5):
    for k, e in envs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) * 1e3 / 200)
for k in shapes:
    print(json.dumps({"blocks": k[0], "threads": k[1], "med_us": round(statistics.median(res[k]), 2), "min_us": round(min(res[k]), 2)}))
