"""fishing-v4 with (r, K, sigma) arrays: do the three parameter streams, allocated separately by torch
(bases spaced by a power of two), collide in the HBM channel hash like the state streams did before the
arena stagger?  Default allocation vs the same arrays carved out of one arena with 12 KiB-staggered starts."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf


def stagger(env):
    N = env.num_envs
    names = [nm for nm in ("_r_arr", "_K_arr", "_sigma_arr") if getattr(env, nm) is not None]
    arena = torch.empty(len(names) * (N * 4 + 12288 * 12 + 256), dtype=torch.uint8, device="cuda")
    off = 12288 * 5
    for k, nm in enumerate(names):
        v = arena[off:off + N * 4].view(torch.float32)
        v.copy_(getattr(env, nm))
        setattr(env, nm, v)
        off = (off + N * 4 + 12288 * (k + 6) + 255) & ~255
    env._cbuf = None
    env._arena2 = arena


for log2n in (21, 22, 24):
    n = 1 << log2n
    ring = torch.empty((4, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((4, n), device="cuda") * 2 - 1)
    res = {"log2n": log2n}
    for tag in ("default", "staggered"):
        env = gf.make("fishing-v4", num_envs=n, seed=1, sigma=torch.full((n,), 0.05), sigma_p=0.1)
        if tag == "staggered":
            stagger(env)
        env.reset(); env.step_many(acts, 50)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_many(acts, 100); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        res[tag + "_us"] = round(statistics.median(ts), 2)
        del env
        torch.cuda.empty_cache()
    res["staggered_TBps_37B"] = round(n * 37 / res["staggered_us"] / 1e6, 2)
    print(json.dumps(res), flush=True)
