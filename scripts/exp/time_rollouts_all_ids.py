"""In-kernel random-policy and escapement rollouts of every id at N = 2^22 (505 steps per launch): env-steps/s by HIP events."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
for k in (0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11):
    idn = "fishing-v%d" % k
    kw = {} if k == 11 else dict(sigma=0.1)
    env = gf.make(idn, num_envs=n, seed=1, track_returns=True, **kw)
    if k == 11:
        for d in env.model_params.values():
            d["sigma"] = 0.1
    env.reset()
    row = {"id": idn}
    for pol, param in (("random", 0.0), ("escapement", 0.5)):
        env.rollout(505, policy=pol, param=param)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.rollout(505, policy=pol, param=param); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        row[pol + "_env_steps_per_s"] = "%.4g" % (n * 505 / statistics.median(ts) * 1e3)
    print(json.dumps(row), flush=True)
    del env
