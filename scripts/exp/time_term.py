"""float32 step with the terminal-observation record (29 B/env-step) or the ballot bitmask (25.125 B): the lean TERM /
BITS instantiations vs the general kernel."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
from gym_fishing_amd import _capi
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
for opt, general in (("term", False), ("term", True), ("bits", False), ("bits", True), ("term", False), ("term", True),
                     ("bits", False), ("bits", True)):
    kw = dict(record_terminal_obs=True) if opt == "term" else dict(done_bits=True)
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, **kw)
    env.reset()
    if general:
        p = env._c_params(); p.flags |= _capi.FLAG_GENERAL_KERNEL
    env.step_many(acts, 200)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, 400); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 2.5)
    us = statistics.median(ts)
    per = 29 if opt == "term" else 25.125
    print(json.dumps({"stream": opt, "kernel": "general" if general else "lean", "us": round(us, 2),
                      "TBps": round(n * per / us / 1e6, 2)}), flush=True)
    del env
