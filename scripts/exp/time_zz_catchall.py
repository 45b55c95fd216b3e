import json, os, sys, torch
sys.path.insert(0, "/root/repo")
import bench, gym_fishing_amd as gf
cfg = bench.CONFIGS["v1"]
for ln, dtype, kw, tag in ((24, torch.float64, {}, "f64 bare"), (24, torch.float64, dict(track_returns=True), "f64 returns"),
                           (26, torch.float32, dict(record_terminal_obs=True), "f32 terminal_obs"),
                           (22, torch.float32, dict(record_terminal_obs=True), "f32 terminal_obs"),
                           (22, torch.float64, {}, "f64 bare")):
    n = 1 << ln
    acts = bench.make_actions(torch, cfg, n, 2)
    us = []
    for rep in range(2):
        env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1234, auto_reset=True, dtype=dtype, **kw)
        env.reset(); env.step_many(acts, 16)
        us.append(min(bench.timed_steps(torch, env, acts, 40, spin_ms=15.0)[0] for _ in range(2)))
        name = env.step_kernel_name(acts[0]); del env; torch.cuda.empty_cache()
    print(json.dumps({"log2_n": ln, "what": tag, "us": [round(u, 1) for u in us], "kernel": name}), flush=True)
    del acts; torch.cuda.empty_cache()
