"""float64 parity layout of the growth zoo at N = 2^22, returns, random policy: us per step (HIP events)."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
acts = torch.empty((8, n + 3072), device="cuda")[:, :n]
acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
for idn in ("fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10", "fishing-v11"):
    kw = {} if idn == "fishing-v11" else dict(sigma=0.1)
    env = gf.make(idn, num_envs=n, seed=1, track_returns=True, dtype=torch.float64, **kw)
    if idn == "fishing-v11":
        for d in env.model_params.values():
            d["sigma"] = 0.1
    env.reset()
    env.step_many(acts, 100)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    byt = 53 + (16 if idn == "fishing-v10" else 0) + (8 if idn == "fishing-v11" else 0)
    us = statistics.median(ts)
    print(json.dumps({"tag": os.environ.get("ZOO_TAG", ""), "id": idn, "us_per_step": round(us, 2), "bytes_per_env_step": byt,
                      "frac_of_8TBps": round(n * byt / us / 8e6, 3), "kernel": env.step_kernel_name()}), flush=True)
    del env
