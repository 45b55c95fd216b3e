#!/usr/bin/env python3
"""Sweep launch shape / workload variants of the step kernel on one GPU (HIP-event timing,
interleaved rounds in one process).  Prints one line per variant: median us/launch, GB/s."""
import itertools
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_fishing_amd as gf  # noqa: E402


def time_variant(env, actions, steps, rounds):
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step_many(actions, steps)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / steps)
    return out


def main():
    log2n = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "22").split(",")]
    variants = []
    for ln in log2n:
        for env_id, sigma, ret, dtype in (("fishing-v1", 0.1, False, torch.float32), ("fishing-v1", 0.0, False, torch.float32),
                                          ("fishing-v1", 0.1, True, torch.float32), ("fishing-v1", 0.1, False, torch.float64),
                                          ("fishing-v0", 0.1, False, torch.float32), ("fishing-v2", 0.1, False, torch.float32),
                                          ("fishing-v4", 0.1, False, torch.float32)):
            for blocks, threads in ((0, 0), (1024, 256), (4096, 256), (2048, 128), (4096, 128), (4096, 64)):
                if (env_id, sigma, ret, dtype) != ("fishing-v1", 0.1, False, torch.float32) and (blocks, threads) != (0, 0):
                    continue
                variants.append((ln, env_id, sigma, ret, dtype, blocks, threads))
    ring = 4
    for ln, env_id, sigma, ret, dtype, blocks, threads in variants:
        n = 1 << ln
        env = gf.make(env_id, sigma=sigma, num_envs=n, seed=1, track_returns=ret, dtype=dtype,
                      launch_blocks=blocks, launch_threads=threads)
        env.reset()
        if env_id == "fishing-v0":
            actions = torch.randint(0, 100, (ring, n), device="cuda", dtype=torch.int32)
        else:
            actions = torch.rand((ring, n), device="cuda") * 2 - 1
        steps = max(20, min(400, int(2e9 / n / 30)))
        time_variant(env, actions, steps, 1)
        ts = time_variant(env, actions, steps, 5)
        per = {"fishing-v4": 37}.get(env_id, 25) + (8 if ret else 0)
        if dtype == torch.float64:
            per = {"fishing-v4": 61}.get(env_id, 37) + (16 if ret else 0)
        med = statistics.median(ts)
        print(json.dumps({"log2n": ln, "id": env_id, "sigma": sigma, "returns": ret, "dtype": str(dtype)[6:],
                          "blocks": blocks, "threads": threads, "us": round(med, 2), "min_us": round(min(ts), 2),
                          "GBps": round(n * per / med / 1e3, 1), "env_steps_per_s": "%.3e" % (n / med * 1e6)}), flush=True)
        del env, actions
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
