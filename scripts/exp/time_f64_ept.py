"""float64 parity layout, round 3: fishing_step_f64 with two envs per thread (the dispatch's pick while a step's streams are
cache-resident) next to four per thread (forced by an explicit workgroup cap, which the dispatch honours by keeping E = 4),
back to back, HIP events.  One JSON line per (N, variant)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for env_id, kw in (("fishing-v1", dict(sigma=0.1)), ("fishing-v2", dict(sigma=0.1)), ("fishing-v4", dict())):
    for ln in (19, 20, 21, 22, 23):
        n = 1 << ln
        ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
        out = {}
        for ret in (False, True):
            for name, extra in (("E2", {}), ("E4", dict(launch_blocks=4096))):
                e = gf.make(env_id, num_envs=n, seed=1, dtype=torch.float64, track_returns=ret, **kw, **extra)
                e.reset()
                K = 200 if ln <= 22 else 60
                v = []
                for rnd in range(3):
                    e.step_many(acts, 30)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); e.step_many(acts, K); e1.record(); torch.cuda.synchronize()
                    v.append(e0.elapsed_time(e1) * 1e3 / K)
                out[name + ("_ret" if ret else "")] = round(statistics.median(v), 2)
                out["kernel_" + name + ("_ret" if ret else "")] = e.step_kernel_name(acts[0])
                del e
        print(json.dumps(dict(env=env_id, log2_n=ln, **out)), flush=True)
        del ring, acts
        torch.cuda.empty_cache()
