"""fishing-v4 (derived parameters, sigma array) at N = 2^24 .. 2^26: workgroup cap of the lean kernel, with returns."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for ln, steps in ((24, 60), (25, 40), (26, 24)):
    n = 1 << ln
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    sig = torch.full((n,), 0.05, device="cuda")
    for ret in (False, True):
        row = {}
        for cap in (768, 1024, 1536, 2048, 4096):
            env = gf.make("fishing-v4", num_envs=n, seed=1, sigma=sig, track_returns=ret, launch_blocks=cap); env.reset()
            env.step_many(acts, steps)
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); env.step_many(acts, steps); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / steps)
            row[cap] = round(statistics.median(ts), 1)
            del env
        B = 37 if ret else 29
        best = min(row, key=row.get)
        print(json.dumps({"log2_n": ln, "returns": ret, "us_by_cap": row, "best_cap": best, "best_frac": round(n * B / row[best] / 8e6, 3),
                          "default_frac": round(n * B / row[4096] / 8e6, 3)}), flush=True)
    del ring, acts, sig; torch.cuda.empty_cache()
