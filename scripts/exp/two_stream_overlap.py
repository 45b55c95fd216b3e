"""Does splitting the env batch over S concurrent streams hide the per-launch ramp/tail (~3 us of a
16 us launch)?  S envs of n/S envs each, every one enqueuing K dependent launches on its own stream."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_fishing_amd as gf
n, K = 1 << 22, 2000
for ret in (False, True):
    for S in (1, 2, 4, 8):
        m = n // S
        envs, acts, streams = [], [], []
        for s in range(S):
            envs.append(gf.make("fishing-v1", sigma=0.1, num_envs=m, seed=1, env_offset=s * m, track_returns=ret))
            ring = torch.empty((8, m + 3072), device="cuda"); a = ring[:, :m]; a.copy_(torch.rand((8, m), device="cuda") * 2 - 1)
            acts.append(a); streams.append(torch.cuda.Stream())
            envs[-1].reset()
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    envs[s].step_many(acts[s], K)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e6)
        print(json.dumps({"returns": ret, "streams": S, "us_per_step_all_envs": round(best, 2),
                          "TBps": round(n * (33 if ret else 25) / best / 1e6, 2)}), flush=True)
        del envs, acts
