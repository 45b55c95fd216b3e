"""Workgroup cap of the lean kernel (grid-stride over 1024-env tiles): bare and returns variants at N = 2^22."""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n = 1 << 22
ring = torch.empty((8, n + 3072), device="cuda"); acts = ring[:, :n]; acts.copy_(torch.rand((8, n), device="cuda") * 2 - 1)
for rnd in range(2):
    for blocks in (512, 1024, 1365, 2048, 2731, 4096):
        res = {"blocks": blocks}
        for ret in (False, True):
            env = gf.make("fishing-v1", sigma=0.1, num_envs=n, seed=1, track_returns=ret, launch_blocks=blocks)
            env.reset(); env.step_many(acts, 300)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); env.step_many(acts, 400); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 2.5)
            res["ret" if ret else "bare"] = round(statistics.median(ts), 2)
            del env
        print(json.dumps(res), flush=True)
