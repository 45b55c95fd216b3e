"""Workgroup cap of the lean step kernel at HBM-resident sizes: N = 2^24 .. 2^27, bare and with the return record."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for ln, steps in ((24, 100), (25, 60), (26, 40), (27, 20)):
    n = 1 << ln
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    for ret in (False, True):
        row = {}
        for cap in (256, 384, 512, 768, 1024, 2048, 4096):
            env = gf.make("fishing-v1", num_envs=n, seed=1, sigma=0.1, track_returns=ret, launch_blocks=cap); env.reset()
            env.step_many(acts, steps)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / steps)
            row[cap] = round(statistics.median(ts), 1)
            del env
        B = 33 if ret else 25
        best = min(row, key=row.get)
        print(json.dumps({"log2_n": ln, "returns": ret, "us_by_cap": row, "best_cap": best,
                          "best_GBps": round(n * B / row[best] / 1e3), "default_GBps": round(n * B / row[4096] / 1e3)}), flush=True)
    del ring, acts; torch.cuda.empty_cache()
