"""fishing-v5..v10 float32 step at N = 2^22: lean kernel vs the general kernel."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
from gym_fishing_amd import _capi
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
for idn in ("fishing-v1", "fishing-v5", "fishing-v6", "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10"):
    res = {"id": idn}
    for general in (False, True):
        env = gf.make(idn, sigma=0.1, num_envs=n, seed=1)
        env.reset()
        if general:       # the env caches its parameter struct: force the general kernel in place
            p = env._c_params(); p.flags |= _capi.FLAG_GENERAL_KERNEL
        env.step_many(acts, 100)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 5)
        res["general_us" if general else "lean_us"] = round(statistics.median(ts), 2)
    res["lean_TBps"] = round(n * (33 if idn == "fishing-v10" else 25) / res["lean_us"] / 1e6, 2)
    print(json.dumps(res), flush=True)
