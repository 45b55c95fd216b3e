"""N = 2^26: exact instantiation with the forward walk vs the catch-all with the zig-zag walk, for requests that have no exact
zig-zag instantiation (fishing-v1 with K = 1.5, the zoo).  Run once with the product library and once with a
-DFISHING_NO_ZOO_HOT build (zoo on its catch-alls)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, gym_fishing_amd as gf
n = 1 << 26
cfg = bench.CONFIGS["v1"]
acts = bench.make_actions(torch, cfg, n, 2)
for idn, kw in (("fishing-v1", dict(K=1.5, init_state=1.1)), ("fishing-v1", dict()), ("fishing-v9", dict()), ("fishing-v6", dict())):
    us = []
    for rep in range(2):
        env = gf.make(idn, sigma=0.1, num_envs=n, seed=1, **kw)
        env.reset(); env.step_many(acts, 16)
        us.append(min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2)))
        name = env.step_kernel_name(acts[0]); del env; torch.cuda.empty_cache()
    print(json.dumps({"lib": os.path.basename(os.environ.get("FISHING_HIP_LIB", "product")), "id": idn, "kw": kw, "us": [round(u, 1) for u in us], "kernel": name}), flush=True)
