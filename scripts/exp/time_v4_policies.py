"""fishing-v4 step time by how often envs finish: the random policy ends ~2/3 of the episodes every
step (each needs a Philox + Box-Muller (K, r) redraw), a gentle policy almost none."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
for tag, lo, width in (("random [-1,1)", -1.0, 2.0), ("gentle [-1,-0.7)", -1.0, 0.3)):
    ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]
    acts.copy_(torch.rand((8, n), device="cuda") * width + lo)
    for idn in ("fishing-v1", "fishing-v4"):
        env = gf.make(idn, sigma=0.05, num_envs=n, seed=1)
        env.reset(); env.step_many(acts, 300)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 200); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 5)
        frac = float(env._done.float().mean())
        print(json.dumps({"id": idn, "policy": tag, "us": round(statistics.median(ts), 2), "done_frac": round(frac, 3)}), flush=True)
        del env
