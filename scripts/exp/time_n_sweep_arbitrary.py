"""Per-step time of fishing-v1 (with the return record) at batch sizes people actually type -- powers of ten, odd sizes --
through step_many (C-enqueued) and through a Python env.step() loop."""
import json, os, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
for n in (1000, 4096, 10_000, 50_000, 50_001, 100_000, 1_000_000, 3_000_000, 10_000_000):
    stride = (n + 3075) // 4 * 4
    ring = torch.empty((8, stride), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
    rows = [acts[k] for k in range(8)]
    env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=True)
    env.reset(); env.step_many(acts, 200)
    ts = []
    K = 400 if n <= 1_000_000 else 100
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step_many(acts, K); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        env.step(rows[k & 7])
    torch.cuda.synchronize()
    py = (time.perf_counter() - t0) / K * 1e6
    us = statistics.median(ts)
    print(json.dumps({"n": n, "padded": bool(env._padded), "step_many_us": round(us, 2), "python_loop_us": round(py, 2),
                      "env_steps_per_s_step_many": "%.3e" % (n / us * 1e6), "kernel": env.step_kernel_name(rows[0])}), flush=True)
    del env, ring, acts, rows
