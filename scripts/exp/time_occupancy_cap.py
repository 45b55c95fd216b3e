"""Experiment: cap the workgroups a CU holds at once with unused dynamic LDS (FISHING_X_DYN_LDS bytes per workgroup;
160 KiB per CU): does running the one-round grids of N = 2^20 .. 2^22 in two or more rounds let one round's compute
overlap the next one's loads?  fishing-v4 (config 5; the VALU-heavy step) and fishing-v1.  Run once per value of the knob."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
lds = int(os.environ.get("FISHING_X_DYN_LDS", "0"))
for env_id, kw in (("fishing-v4", dict(sigma_p=0.1)), ("fishing-v1", dict(sigma=0.1))):
    for ln in (20, 21, 22):
        n = 1 << ln
        if env_id == "fishing-v4":
            kw["sigma"] = torch.full((n,), 0.05, device="cuda")
        ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
        e = gf.make(env_id, num_envs=n, seed=1, track_returns=True, **kw)
        e.reset()
        v = []
        for rnd in range(3):
            e.step_many(acts, 100)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e.step_many(acts, 400); e1.record(); torch.cuda.synchronize()
            v.append(e0.elapsed_time(e1) * 1e3 / 400)
        print(json.dumps(dict(dyn_lds=lds, wg_per_cu=(160 * 1024 // (lds + 512)) if lds else 8, env=env_id, log2_n=ln, us=round(statistics.median(v), 2))), flush=True)
        del e, ring, acts
